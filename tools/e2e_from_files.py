"""End to end from file bytes in host memory to RGB in HBM (mjx_decode_batch: parse on host threads, upload, decode):
python tools/e2e_from_files.py [files] [threads] [de-stuffing: 0 = host, 1 = device; left out = the library's choice, MJX_DESTUFF_AUTO]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
mjx = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dd = bool(int(sys.argv[3])) if len(sys.argv) > 3 else None
ctx = mjx.Context(0)
uniq = [mjx.synth_jpeg(3840, 2160, "420", 75, seed=s) for s in range(64)]
datas = [uniq[i % 64] for i in range(n)]
b, st = mjx.decode_batch(ctx, datas[:8], threads=threads, device_destuff=dd); b.close()          # warm-up (first use of the device)
for rep in range(3):
    t = time.perf_counter()
    b, st = mjx.decode_batch(ctx, datas, threads=threads, device_destuff=dd)
    dt = time.perf_counter() - t
    assert all(s == mjx.OK for s in st)
    print("%d 4K files (%.0f MB), device_destuff=%s, threads=%d: %.1f ms = %.1f Gpx/s, %.0f files/s" % (n, sum(map(len, datas)) / 1e6, dd, threads, dt * 1e3, n * 3840 * 2160 / dt / 1e9, n / dt))
    b.close()
