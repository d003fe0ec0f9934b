"""First-contact probe: decode the fixtures on the GPU and print how far the result is from the oracle."""
import os, sys, time, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build()
mjx = ge.load_package()
import oracle_binding as orc
ctx = mjx.Context(0, profiling=True)
cases = [(n, open(os.path.join(ROOT, "tests", "data", n), "rb").read()) for n in ["huff_simple0.jpg", "lena-bw.jpeg", "lena.jpeg", "2x2-chroma.jpeg"]]
cases += [("synth %dx%d %s" % (w, h, s), mjx.synth_jpeg(w, h, s, 75, seed=3)) for (w, h, s) in [(64, 48, "444"), (61, 45, "420"), (1920, 1080, "420"), (3840, 2160, "420")]]
for name, data in cases:
    try:
        t = time.time()
        scan = mjx.ParsedScan(data)
        b = mjx.Batch(ctx, [scan], keep_coefs=True)
        b.decode(); b.wait()
        dt = time.time() - t
        ref = orc.decode(data, layout=orc.LAYOUT_STD)
        co = b.coefs(0); rgb = b.rgb(0)
        want = orc.interleave(ref)
        t0 = np.array_equal(co, want)
        nbad = int((co != want).any(axis=1).sum())
        first_bad = int(np.argmax((co != want).any(axis=1))) if nbad else -1
        d = np.abs(rgb.astype(int) - ref.rgb.astype(int))
        print("%-28s status %d T0 %s (bad blocks %d/%d first %d) rgb maxdiff %d frac>0 %.5f frac>1 %.5f  %.1f ms  %s" % (
            name, b.status(0), "EQUAL" if t0 else "DIFF", nbad, len(want), first_bad, d.max(), (d > 0).mean(), (d > 1).mean(), dt * 1e3,
            {k: round(v[0], 3) for k, v in b.kernel_ms().items()}))
        b.close()
    except Exception:
        traceback.print_exc()
