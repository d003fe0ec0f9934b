import faulthandler, os, sys, time
faulthandler.dump_traceback_later(40, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
mjx = ge.load_package()
keep = int(sys.argv[1]) if len(sys.argv) > 1 else 1
times = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = mjx.Context(0)
t = time.time()
datas = mjx.synth_batch(64, 1920, 1080, "420", 75)
print("synth", time.time() - t, flush=True)
base = mjx.Batch(ctx, [mjx.ParsedScan(d) for d in datas], keep_coefs=bool(keep))
base.decode(); base.wait(); print("base ok", time.time() - t, flush=True)
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
big = base.tile(times); print("tiled", len(big), time.time() - t, flush=True)
big.decode(); print("enqueued", time.time() - t, flush=True)
big.wait(); print("waited", time.time() - t, flush=True)
mx, cnt = big.compare_rgb(list(range(len(big))), base, [i % 64 for i in range(len(big))])
print("compare", int(mx.max()), time.time() - t, flush=True)
