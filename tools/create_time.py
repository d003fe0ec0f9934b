"""Cost of mjx_batch_create (planning, decode tables, upload, work buffers) for batches of 4K files: python tools/create_time.py"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
mjx = ge.load_package()
ctx = mjx.Context(0)
datas = [mjx.synth_jpeg(3840, 2160, '420', 75, seed=s) for s in range(64)]
scans = [mjx.ParsedScan(d) for d in datas]
for n in (1, 16, 64, 64, 256):
    ss = (scans * 4)[:n]
    t = time.perf_counter(); b = mjx.Batch(ctx, ss); dt = time.perf_counter() - t
    print(n, "files: create %.1f ms total, %.3f ms per file" % (dt * 1e3, dt * 1e3 / n))
    b.close()
