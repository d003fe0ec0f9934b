"""BASELINE.json configs 1-3 on one GPU: the reference's sample files one at a time (latency, not throughput).
Prints per file: device-resident decode time (mjx_batch_decode + wait, median of 50), one-shot mjx_decode (parse + upload +
decode + copy back, median of 20), oracle time on one core, and the parity result against the oracle."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
import oracle_binding as orc
mjx = ge.load_package()
ctx = mjx.Context(0)
for name, layout in [("huff_simple0.jpg", "std"), ("lena-bw.jpeg", "std"), ("lena.jpeg", "std"), ("2x2-chroma.jpeg", "ref")]:
    data = open(os.path.join(ROOT, "tests", "data", name), "rb").read()
    lay = mjx.LAYOUT_REF_COMPAT if layout == "ref" else mjx.LAYOUT_STANDARD
    b = mjx.Batch(ctx, [mjx.ParsedScan(data)], keep_coefs=True, layout=lay)
    ts = []
    for _ in range(50):
        t = time.perf_counter(); b.decode(); b.wait(); ts.append(time.perf_counter() - t)
    t = time.perf_counter()
    ref = orc.decode(data, layout=orc.LAYOUT_REF if layout == "ref" else orc.LAYOUT_STD, faithful_cos=True, faithful_huff=True)
    t_cpu = time.perf_counter() - t
    t0 = bool(np.array_equal(b.coefs(0), orc.interleave(ref)))
    d = int(np.abs(b.rgb(0).astype(int) - ref.rgb.astype(int)).max())
    b.close()
    os_ = []
    for _ in range(20):
        t = time.perf_counter(); mjx.decode(data, layout=lay); os_.append(time.perf_counter() - t)
    px = ref.rgb.shape[0] * ref.rgb.shape[1]
    print("%-18s %4dx%-4d layout=%s  resident decode %.3f ms (%.1f Mpx/s)  one-shot mjx_decode %.3f ms  reference algorithm on 1 core %.1f ms (%.2f Mpx/s)  T0 equal %s, max |RGB diff| %d"
          % (name, ref.rgb.shape[1], ref.rgb.shape[0], layout, statistics.median(ts) * 1e3, px / statistics.median(ts) / 1e6,
             statistics.median(os_) * 1e3, t_cpu * 1e3, px / t_cpu / 1e6, t0, d))
