"""Randomised check of multi-scan pictures read from their scans' streams (DevImage::planar, round 5) on a GPU box:
    python tools/fuzz_planar.py [seed] [cases]
Random sizes (weighted towards MCU rows a little shorter / longer than a stage-B tile), 4:2:0 / 4:2:2 / 4:4:4, both scan forms,
restart intervals; every twin is decoded alone and as 48 / 160 tiled copies (workgroups that walk several tiles), with and without
kept coefficients: the same bytes, the source's picture bit for bit."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import __graft_entry__ as ge, oracle_binding as orc, make_multiscan
mjx = ge.load_package(); orc.lib()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(seed)
ctx = mjx.Context(0)
direct = 0
for k in range(cases):
    sub = ["420", "420", "422", "444"][int(rng.integers(0, 4))]
    mw = {"420": 16, "422": 16, "444": 8}[sub]
    tile = {"420": 32, "422": 32, "444": 64}[sub]
    mcux = int(rng.choice([tile - 2, tile - 1, tile, tile + 1, tile + 2, 2 * tile - 1, 2 * tile + 1, int(rng.integers(1, 3 * tile)), int(rng.integers(3 * tile, 9 * tile))]))
    w = max(1, mcux * mw - int(rng.integers(0, mw)))
    h = int(rng.integers(1, 400)) if rng.random() < 0.8 else int(rng.integers(400, 1400))
    q = int(rng.integers(20, 98))
    src = mjx.synth_jpeg(w, h, sub, q, seed=int(rng.integers(0, 1 << 30)), noise_sigma=float(rng.uniform(0, 20)))
    kw = dict(chroma_together=bool(rng.integers(0, 2)))
    if rng.random() < 0.3: kw["restart"] = int(rng.integers(1, 80))
    tw = make_multiscan.twin(src, **kw)
    ref = orc.decode(src, layout=orc.LAYOUT_STD)
    want = None
    for copies in (1, 48, 160):
        outs = []
        for keep in (True, False):
            base = mjx.Batch(ctx, [mjx.ParsedScan(tw), mjx.ParsedScan(src)], keep_coefs=keep)
            b = base.tile(copies) if copies > 1 else base
            b.decode(); b.wait()
            assert all(b.status(i) == mjx.OK for i in range(len(b))), (k, w, h, sub, kw, copies, keep)
            if not keep and copies == 1:
                try:
                    b.coefs(0)
                except Exception:
                    direct += 1
            outs.append([b.rgb(i) for i in (0, 1, len(b) - 2, len(b) - 1)])
            b.close()
            if b is not base: base.close()
        if want is None: want = outs[0][1]                     # the interleaved source's picture
        for o in outs:
            for x in o:
                assert np.array_equal(x, want), ("differs", k, w, h, sub, q, kw, copies)
    assert np.abs(want.astype(int) - ref.rgb.astype(int)).max() <= 1, (k, w, h, sub)
print("fuzz_planar ok: seed %d, %d twins (%d took the direct path), alone / x48 / x160, with and without kept coefficients" % (seed, cases, direct))
