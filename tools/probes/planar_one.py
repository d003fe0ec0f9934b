import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as ge, oracle_binding as orc
mjx = ge.load_package()
d = open(sys.argv[1], "rb").read()
ctx = mjx.Context(0)
ref = orc.decode(d, layout=orc.LAYOUT_STD, ext_dri=True, ext_1bit=True, ext_multiscan=True)
for keep in (True, False):
    b = mjx.Batch(ctx, [mjx.ParsedScan(d)], keep_coefs=keep)
    b.decode(); b.wait()
    rgb = b.rgb(0)
    diff = np.abs(rgb.astype(int) - ref.rgb.astype(int))
    ys, xs = np.nonzero(diff.max(axis=2) > 1)
    print("keep", keep, "status", b.status(0), "max diff", diff.max(), "bad pixels", len(ys), "bbox", (ys.min(), ys.max(), xs.min(), xs.max()) if len(ys) else None)
    if len(ys):
        # which 16x16 MCUs
        m = sorted(set(zip((ys // 16).tolist(), (xs // 16).tolist())))
        print(" bad MCUs (row, col):", m[:40], len(m))
    b.close()
