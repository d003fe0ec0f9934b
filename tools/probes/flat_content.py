"""Round 5 probe: pictures with large flat areas (documents, graphics).  Huffman decoding from a guessed state synchronises through
CONTENT; a run of identical flat MCUs is periodic, and a decode that enters it out of phase can stay on a self-consistent wrong parse
until the run ends -- the subsequences inside it are then re-decoded one after the other.  Times and parity of such
pictures, one at a time and as a batch:   python tools/probes/flat_content.py"""
import io, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from PIL import Image
import __graft_entry__ as ge, oracle_binding as orc
mjx = ge.load_package(); orc.lib()
rng = np.random.default_rng(1)


def picture(kind, w=3840, h=2160):
    a = np.full((h, w, 3), 255, np.uint8)
    if kind == "white page + noisy lines":
        for y in range(100, h - 100, 60):
            a[y:y + 20, 200:w - 200] = rng.integers(0, 255, (20, w - 400, 1))
    elif kind == "flat halves":
        a[:, :w // 2] = 128
    elif kind == "flat quarters":
        a[:h // 2, :w // 2] = (200, 30, 30); a[h // 2:, w // 2:] = (30, 30, 200)
    elif kind == "photo-like":
        return mjx.synth_jpeg(w, h, "420", 75, seed=1)
    buf = io.BytesIO(); Image.fromarray(a).save(buf, "JPEG", quality=75, subsampling=2)
    return buf.getvalue()


ctx = mjx.Context(0, profiling=True)
for kind in ("photo-like", "all white", "white page + noisy lines", "flat halves", "flat quarters"):
    d = picture(kind)
    ref = orc.decode(d, layout=orc.LAYOUT_STD)
    for copies in (1, 64):
        base = mjx.Batch(ctx, [mjx.ParsedScan(d)], keep_coefs=copies == 1)
        b = base.tile(copies) if copies > 1 else base
        b.decode(); b.wait()
        t0 = time.perf_counter(); b.decode(); b.wait(); el = time.perf_counter() - t0
        ok = all(b.status(i) == mjx.OK for i in range(len(b)))
        diff = int(np.abs(b.rgb(len(b) - 1).astype(int) - ref.rgb.astype(int)).max()) if ok else -1
        t0eq = bool(np.array_equal(b.coefs(0), orc.interleave(ref))) if (ok and copies == 1) else None
        print("%-26s %8d bytes  x%-3d %9.3f ms  statuses ok %s  T0 equal %s  max |RGB diff| %d  subsequences %d" %
              (kind, len(d), copies, el * 1e3, ok, t0eq, diff, b.geometry()["subsequences"]), flush=True)
        b.close()
        if b is not base: base.close()
