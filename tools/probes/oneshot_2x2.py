import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
mjx = ge.load_package()
data = open(os.path.join(ROOT, "tests", "data", sys.argv[1] if len(sys.argv) > 1 else "2x2-chroma.jpeg"), "rb").read()
ts = []
for i in range(12):
    t = time.perf_counter(); mjx.decode(data); ts.append(time.perf_counter() - t)
print(" ".join("%.2f" % (x * 1e3) for x in ts))
