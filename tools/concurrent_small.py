"""Several processes decode single pictures on one device at the same time (every decode runs k_huff_merge_loop, whose
workgroups wait for one another): python tools/concurrent_small.py [processes] [decodes]
Checks every result against the first one and reports the slowest decode -- a loop that could not assemble would show up as
a decode of several seconds (it gives up and falls back), a wrong barrier as differing bytes."""
import os, sys, time, hashlib, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    mjx = ge.load_package()
    n = int(sys.argv[2])
    datas = [open(os.path.join(ROOT, "tests", "data", "lena.jpeg"), "rb").read(), mjx.synth_jpeg(3840, 2160, "420", 75, seed=3),
             mjx.synth_jpeg(1920, 1080, "420", 90, seed=5)]
    ref = [None] * len(datas)
    worst = 0.0
    for i in range(n):
        k = i % len(datas)
        t = time.perf_counter()
        rgb = mjx.decode(datas[k])
        worst = max(worst, time.perf_counter() - t)
        h = hashlib.sha256(rgb.tobytes()).hexdigest()
        if ref[k] is None:
            ref[k] = h
        assert ref[k] == h, (i, k)
    print("worker ok: %d decodes, slowest %.2f ms, hashes %s" % (n, worst * 1e3, " ".join(x[:12] for x in ref)))
    sys.exit(0)
procs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(n)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(procs)]
outs = [p.communicate(timeout=900)[0].strip().splitlines()[-1] for p in ps]
for o in outs:
    print(o)
assert all(p.returncode == 0 for p in ps)
assert len({o.split("hashes")[1] for o in outs}) == 1
print("concurrent ok: %d processes" % procs)
