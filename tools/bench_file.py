"""Throughput of the decode path on copies of given JPEG files (device-resident in, device-resident out), e.g. the
restart-interval fixtures:  python tools/bench_file.py tests/golden/pil/dri_420_720p_rows.jpg --copies 8192
Prints one JSON line (Mpixels/s, ms per step, per-kernel ms per step).  Not the contract bench (see bench.py)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

ap = argparse.ArgumentParser()
ap.add_argument("files", nargs="+")
ap.add_argument("--copies", type=int, default=4096)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--chunk-images", type=int, default=0)
a = ap.parse_args()
mjx = ge.load_package()
ctx = mjx.Context(0, profiling=True, throughput_plan=True)
datas = [open(f, "rb").read() for f in a.files]
base = mjx.Batch(ctx, [mjx.ParsedScan(d) for d in datas], chunk_images=a.chunk_images)
assert all(s == mjx.OK for s in base.create_status), base.create_status
reps = max(1, a.copies // len(datas))
batch = base.tile(reps) if reps > 1 else base
for _ in range(a.warmup):
    batch.decode(); batch.wait()
batch.kernel_ms(reset=True)
t0 = time.perf_counter()
for _ in range(a.steps):
    batch.decode()
batch.wait()
el = time.perf_counter() - t0
assert all(batch.status(i) == mjx.OK for i in range(len(batch)))
by, kms = batch.bytes(), batch.kernel_ms()
print(json.dumps({"files": [os.path.basename(f) for f in a.files], "images": len(batch), "Mpixels/s": round(by["pixels"] * a.steps / el / 1e6, 1),
                  "ms_per_step": round(el / a.steps * 1e3, 3), "scan_MB": round(by["scan"] / 1e6, 1),
                  "kernels_ms_per_step": {k: round(v[0] / a.steps, 3) for k, v in kms.items() if v[1]}}))
