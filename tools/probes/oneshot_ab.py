"""Round 5 probe: one-shot mjx_decode and resident decode of the sample files, for an A/B of two builds (MJX_LIB)."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
mjx = ge.load_package()
ctx = mjx.Context(0)
for name in ("lena-bw.jpeg", "lena.jpeg", "2x2-chroma.jpeg"):
    data = open(os.path.join(ROOT, "tests", "data", name), "rb").read()
    for _ in range(10): mjx.decode(data)
    ts = []
    for _ in range(60):
        t = time.perf_counter(); mjx.decode(data); ts.append(time.perf_counter() - t)
    b = mjx.Batch(ctx, [mjx.ParsedScan(data)])
    rs = []
    for _ in range(60):
        t = time.perf_counter(); b.decode(); b.wait(); rs.append(time.perf_counter() - t)
    b.close()
    print("%-16s one-shot median %.3f ms (min %.3f)   resident %.3f ms" % (name, statistics.median(ts) * 1e3, min(ts) * 1e3, statistics.median(rs) * 1e3))
