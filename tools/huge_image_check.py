"""Extreme geometries (192 MP, 65500 pixels wide / tall) against the oracle: python tools/huge_image_check.py"""
import sys, os, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge, oracle_binding as orc
mjx = ge.load_package(); orc.lib()
ctx = mjx.Context(0)
for (w, h, sub, q) in [(16000, 12000, "420", 75), (65500, 600, "444", 50), (600, 65500, "422", 90)]:
    t = time.time(); d = mjx.synth_jpeg(w, h, sub, q, seed=5); print((w, h, sub), "bytes", len(d), "gen %.1fs" % (time.time() - t), flush=True)
    b = mjx.Batch(ctx, [mjx.ParsedScan(d)], keep_coefs=True)
    t = time.time(); b.decode(); b.wait(); print("  gpu %.3fs status %d" % (time.time() - t, b.status(0)), flush=True)
    t = time.time(); ref = orc.decode(d, layout=orc.LAYOUT_STD); print("  oracle %.1fs" % (time.time() - t), flush=True)
    assert np.array_equal(b.coefs(0), orc.interleave(ref))
    diff = np.abs(b.rgb(0).astype(np.int16) - ref.rgb.astype(np.int16)); assert diff.max() <= 1
    print("  ok, off-by-one fraction %.2e" % float((diff > 0).mean()), flush=True)
    b.close()
