"""How the CPU oracle scales with threads on this box (for sizing bench.py's cpu_baseline leg)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import oracle_binding as orc
mjx = ge.load_package()
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try:
        print(f, open(f).read().strip())
    except OSError:
        pass
os.system("grep -m1 'model name' /proc/cpuinfo; free -g | head -2; nproc")
datas = [mjx.synth_jpeg(3840, 2160, "420", 75, s) for s in range(8)]
for t in [int(a) for a in sys.argv[1:]] or [1, 16, 32, 64, 128]:
    sample = [datas[i % 8] for i in range(t)]
    t0 = time.perf_counter()
    px, st = orc.decode_many(sample, t, layout=orc.LAYOUT_REF, faithful=True)
    dt = time.perf_counter() - t0
    print("threads %4d: %.1f s, %.2f Mpx/s (%.2f per thread)" % (t, dt, px / dt / 1e6, px / dt / 1e6 / t), flush=True)
