"""Round 5 probe: one-shot mjx_decode of a 4K picture (24.9 MB of RGB back to the host), call by call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
mjx = ge.load_package()
data = mjx.synth_jpeg(3840, 2160, "420", 75, seed=1)
ts = []
for i in range(10):
    t = time.perf_counter(); img = mjx.decode(data); ts.append(time.perf_counter() - t)
print("4K one-shot", " ".join("%.2f" % (x * 1e3) for x in ts))
