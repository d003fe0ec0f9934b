"""Where the time of a one-shot decode of one file goes (medians of 30): python tools/oneshot_breakdown.py [file]"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
mjx = ge.load_package()
ctx = mjx.Context(0)
name = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "data", "lena.jpeg")
data = open(name, "rb").read()
T = {k: [] for k in ("parse", "create", "decode+wait", "copy rgb", "close", "mjx_decode")}
for it in range(35):
    t0 = time.perf_counter(); s = mjx.ParsedScan(data)
    t1 = time.perf_counter(); b = mjx.Batch(ctx, [s])
    t2 = time.perf_counter(); b.decode(); b.wait()
    t3 = time.perf_counter(); rgb = b.rgb(0)
    t4 = time.perf_counter(); b.close(); s.close()
    t5 = time.perf_counter(); mjx.decode(data)
    t6 = time.perf_counter()
    if it >= 5:
        for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
            T[k].append(v)
print(os.path.basename(name), {k: "%.0f us" % (statistics.median(v) * 1e6) for k, v in T.items()})
