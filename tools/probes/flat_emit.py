"""Round 5 probe: large pictures (cut into long subsequences: single decode, one recorded decode per subsequence) with large flat areas."""
import io, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from PIL import Image
import __graft_entry__ as ge, oracle_binding as orc
mjx = ge.load_package(); orc.lib()
rng = np.random.default_rng(2)
ctx = mjx.Context(0, profiling=True)
for name in ("noise", "left half grey, right half noise", "grey / white columns beside noise"):
    w, h = 7680, 4320
    a = rng.integers(0, 255, (h, w, 3)).astype(np.uint8)
    if name.startswith("left"): a[:, :w // 2] = 128
    if name.startswith("grey /"):
        a[:, :w // 2] = 255
        a[:, :w // 4] = 128
    buf = io.BytesIO(); Image.fromarray(a).save(buf, "JPEG", quality=60, subsampling=2); d = buf.getvalue()
    b = mjx.Batch(ctx, [mjx.ParsedScan(d)], keep_coefs=True)
    b.decode(); b.wait(); b.kernel_ms(reset=True)
    t = time.perf_counter(); b.decode(); b.wait(); el = time.perf_counter() - t
    k = b.kernel_ms()
    ref = orc.decode(d, layout=orc.LAYOUT_STD)
    ok = b.status(0) == 0 and np.array_equal(b.coefs(0), orc.interleave(ref))
    print("%-36s %9d bytes %8.2f ms  T0 equal %s  %s  %s" % (name, len(d), el * 1e3, ok, {n: round(v[0], 2) for n, v in k.items() if v[1]}, b.geometry()), flush=True)
    b.close()
