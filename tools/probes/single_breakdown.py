"""Round 5 probe: per-kernel HIP-event times of one resident decode of the reference's sample files (profiling context), both layouts for 2x2-chroma."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
mjx = ge.load_package()
ctx = mjx.Context(0, profiling=True)
for name, layout in [("lena.jpeg", "std"), ("2x2-chroma.jpeg", "std"), ("2x2-chroma.jpeg", "ref")]:
    data = open(os.path.join(ROOT, "tests", "data", name), "rb").read()
    b = mjx.Batch(ctx, [mjx.ParsedScan(data)], layout=mjx.LAYOUT_REF_COMPAT if layout == "ref" else mjx.LAYOUT_STANDARD)
    for _ in range(5): b.decode(); b.wait()
    b.kernel_ms(reset=True)
    ts = []
    for _ in range(20):
        t = time.perf_counter(); b.decode(); b.wait(); ts.append(time.perf_counter() - t)
    k = b.kernel_ms()
    print(name, layout, "median %.3f ms" % (statistics.median(ts) * 1e3), {n: round(v[0] / 20 * 1e3, 1) for n, v in k.items() if v[1]}, "us per decode;", b.geometry())
    b.close()
