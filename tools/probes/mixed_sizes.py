"""Round 5 probe: a batch of small pictures with and without one very large picture among them (the entropy grids are
pictures x the longest scan's workgroups: the small pictures' empty slots are launched and leave at once)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
mjx = ge.load_package()
ctx = mjx.Context(0, profiling=True, throughput_plan=True)
small = [mjx.synth_jpeg(512, 512, "420", 75, seed=s) for s in range(32)]
big = mjx.synth_jpeg(7680, 4320, "420", 90, seed=99)
for name, extra in (("8192 x 512x512", []), ("8192 x 512x512 + one 7680x4320 q90", [big])):
    base = mjx.Batch(ctx, [mjx.ParsedScan(d) for d in small])
    b = base.tile(256)
    if extra:
        # (one batch: the tiled small pictures cannot be joined with another picture, so build the list explicitly -- fewer copies)
        b.close(); base.close()
        b = mjx.Batch(ctx, [mjx.ParsedScan(d) for d in (small * 64 + extra)])
    b.decode(); b.wait()
    b.kernel_ms(reset=True)
    t = time.perf_counter()
    for _ in range(5): b.decode()
    b.wait()
    el = (time.perf_counter() - t) / 5
    k = b.kernel_ms()
    print(name, "images", len(b), "%.3f ms per step" % (el * 1e3), "unconverged", b.unconverged_runs(), {n: round(v[0] / 5, 2) for n, v in k.items() if v[1]}, b.geometry())
    b.close()
base = mjx.Batch(ctx, [mjx.ParsedScan(d) for d in small * 64])
base.decode(); base.wait(); base.kernel_ms(reset=True)
t = time.perf_counter()
for _ in range(5): base.decode()
base.wait()
print("2048 x 512x512 (the same list without the large one)", "%.3f ms per step" % ((time.perf_counter() - t) / 5 * 1e3), {n: round(v[0] / 5, 2) for n, v in base.kernel_ms().items() if v[1]})
