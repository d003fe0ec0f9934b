"""A multi-scan 4K picture whose scans are long enough for long subsequences (and take the packed stream + the gather kernels),
with and without restart intervals, against the oracle: python tools/big_multiscan_check.py   (GPU box; the twin is encoded in Python: ~1 min)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import __graft_entry__ as ge, oracle_binding as orc, make_multiscan
mjx = ge.load_package(); orc.lib()
ctx = mjx.Context(0, throughput_plan=True)
src = mjx.synth_jpeg(3840, 2160, "420", 92, seed=3, noise_sigma=10.0)
ref = orc.decode(src, layout=orc.LAYOUT_STD)
for kw in (dict(), dict(chroma_together=True), dict(restart=240)):
    tw = make_multiscan.twin(src, **kw)
    datas = [tw, src, tw]
    b = mjx.Batch(ctx, [mjx.ParsedScan(d) for d in datas], keep_coefs=True)
    b.decode(); b.wait()
    for i in range(3):
        assert b.status(i) == mjx.OK, (kw, i, b.status(i))
        assert np.array_equal(b.coefs(i), orc.interleave(ref)), (kw, i)
        assert np.abs(b.rgb(i).astype(int) - ref.rgb.astype(int)).max() <= 1, (kw, i)
    # ... and without kept coefficients, where stage B reads the twin straight from its scans' streams (DevImage::planar): the same bytes
    p = mjx.Batch(ctx, [mjx.ParsedScan(d) for d in datas])
    p.decode(); p.wait()
    for i in range(3):
        assert p.status(i) == mjx.OK and np.array_equal(p.rgb(i), b.rgb(i)), (kw, i)
    print("ok", kw, "file bytes", len(tw), b.geometry())
    p.close()
    b.close()
