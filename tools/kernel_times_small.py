"""Per-kernel time of one resident decode of a single picture (latency view): python tools/kernel_times_small.py"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
mjx = ge.load_package()
ctx = mjx.Context(0, profiling=True)
cases = [(n, open(os.path.join(ROOT, "tests", "data", n), "rb").read()) for n in ["lena.jpeg", "lena-bw.jpeg", "2x2-chroma.jpeg", "huff_simple0.jpg"]]
cases.append(("synthetic 4K q75", mjx.synth_jpeg(3840, 2160, "420", 75, seed=3)))
cases.append(("synthetic 1080p q75", mjx.synth_jpeg(1920, 1080, "420", 75, seed=3)))
for name, data in cases:
    b = mjx.Batch(ctx, [mjx.ParsedScan(data)])
    for _ in range(5): b.decode(); b.wait()
    b.kernel_ms(reset=True)
    ts = []
    for _ in range(20):
        t = time.perf_counter(); b.decode(); b.wait(); ts.append(time.perf_counter() - t)
    k = b.kernel_ms()
    print("%-20s %.0f us wall;" % (name, statistics.median(ts) * 1e6), {n: (round(v[0] / 20 * 1000, 1), v[1] // 20) for n, v in k.items() if v[1]}, "us per decode", b.geometry())
    b.close()
