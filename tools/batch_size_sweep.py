"""Resident decode time against batch size (4K 4:2:0 q75 pictures): python tools/batch_size_sweep.py
Shows where the small-batch path (short subsequences, merge rounds in one launch) hands over to the throughput path."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
mjx = ge.load_package()
ctx = mjx.Context(0)
uniq = [mjx.synth_jpeg(3840, 2160, "420", 75, seed=s) for s in range(64)]
scans = [mjx.ParsedScan(d) for d in uniq]
for n in (1, 2, 4, 8, 12, 16, 24, 32, 64, 128, 256, 512):
    b = mjx.Batch(ctx, [scans[i % 64] for i in range(n)])
    for _ in range(3): b.decode(); b.wait()
    ts = []
    for _ in range(15):
        t = time.perf_counter(); b.decode(); b.wait(); ts.append(time.perf_counter() - t)
    g = b.geometry()
    dt = statistics.median(ts)
    print("%4d pictures: %7.3f ms  %6.1f Gpx/s  (%d subsequences, %d chunk(s))" % (n, dt * 1e3, n * 3840 * 2160 / dt / 1e9, g["subsequences"], g["chunks"]))
    b.close()
