"""Round 6: randomised parity check of the single decode in 256-lane workgroups (mjx_plan.cpp: a scan that fills ONE 256-lane workgroup
with long subsequences takes them when its whole batch does).  tools/fuzz_parity.py never reaches that rule -- its pictures are at most
900 x 700 --, so this draws Pillow-encoded pictures whose de-stuffed scan lies inside the rule's range (1.835 .. 2.62 Mbit): random size
around 1080p, sampling, quality, standard or optimised tables, smooth / noisy / mixed content, a flat half now and then; batches them
(throughput plan: the cut of a batch that fills the device), tiles them over several chunks, and checks every picture against the CPU
oracle: coefficients bit-exact (T0), RGB within 1, the emitting kernel must have run, tiled copies equal to their originals.
    python tools/probes/fuzz_long_fit.py [seed] [batches] [pictures per batch]
"""
import io, os, sys
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import oracle_binding as orc

mjx = ge.load_package()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 3
per = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rng = np.random.default_rng(seed)
ctx = mjx.Context(0, profiling=True, throughput_plan=True)
LO, HI = 256 * 8192 * 7 // 8, 256 * 10240


def picture(w, h):
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100 * np.sin(xx / rng.uniform(5, 120) + yy / rng.uniform(5, 120) + p) for p in (0.0, 2.0, 4.0)], -1)
    kind = int(rng.integers(0, 3))
    img = base + rng.normal(0, rng.uniform(1, 25), (h, w, 3)) if kind else base + rng.normal(0, 4, (h, w, 3))
    if kind == 2:
        img[:, : w // 2] = rng.integers(0, 256, 3)            # a flat half: long runs of identical MCUs
    return np.clip(img, 0, 255).astype(np.uint8)


def draw():
    for _ in range(200):
        w, h = int(rng.integers(1200, 2300)), int(rng.integers(800, 1400))
        arr = picture(w, h)
        kw = dict(quality=int(rng.integers(40, 96)), optimize=bool(rng.random() < 0.5), subsampling=int(rng.integers(0, 3)))
        for _ in range(6):                                     # steer the quality until the scan lands in the range
            buf = io.BytesIO()
            try:
                Image.fromarray(arr).save(buf, "JPEG", **kw)
            except OSError:                                    # (Pillow gives up on some option combinations: another picture)
                break
            d = buf.getvalue()
            sc = mjx.ParsedScan(d)
            bits = sc.desc.scan_len * 8
            sc.close()
            if LO <= bits <= HI:
                return d, (w, h, kw, bits)
            kw["quality"] = int(np.clip(kw["quality"] + (8 if bits < LO else -8), 5, 98))
    raise SystemExit("no picture in range")


bad = 0
for bi in range(batches):
    items = [draw() for _ in range(per)]
    datas = [x[0] for x in items]
    refs = [orc.decode(d, layout=orc.LAYOUT_STD, ext_1bit=True) for d in datas]       # (optimised tables may hold 1-bit codes: SURVEY Q8)
    b = mjx.Batch(ctx, [mjx.ParsedScan(d) for d in datas], keep_coefs=True, chunk_images=int(rng.integers(0, per + 1)))
    b.kernel_ms(reset=True)
    b.decode(); b.wait(); b.decode(); b.wait()
    k = b.kernel_ms()
    if not k["huff_emit"][1]:
        print("batch", bi, "did not take the emitting kernel", k); bad += 1
    for i, ref in enumerate(refs):
        ok = b.status(i) == mjx.OK and np.array_equal(b.coefs(i), orc.interleave(ref)) and np.abs(b.rgb(i).astype(int) - ref.rgb.astype(int)).max() <= 1
        if not ok:
            print("MISMATCH batch", bi, "picture", i, items[i][1], "status", b.status(i)); bad += 1
    t = b.tile(int(rng.integers(3, 30)))
    t.decode(); t.wait()
    n = len(datas)
    mx, cnt = t.compare_rgb(list(range(n, len(t))), t, [i % n for i in range(n, len(t))])
    if int(mx.max()) != 0 or not all(t.status(i) == mjx.OK for i in range(len(t))):
        print("tiled copies differ in batch", bi, int(mx.max())); bad += 1
    print("batch", bi, ":", n, "pictures,", len(t), "tiled, unconverged runs", b.unconverged_runs(), "fall-backs seen" if k["huff_write"][1] else "", [x[1][3] for x in items][:4])
    t.close(); b.close()
print("fuzz_long_fit seed", seed, ":", batches * per, "pictures,", bad, "failures")
sys.exit(1 if bad else 0)
