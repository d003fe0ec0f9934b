"""Throughput of multi-scan files against their interleaved twin: a synthetic 4K 4:2:0 picture, the same
coefficients as one scan per component and as luma + interleaved chroma (tests/golden/make_multiscan.py; encoding them in
Python takes about a minute), N copies each, device-resident in and out.  One JSON line per form.
    python tools/bench_multiscan.py [--copies 1024] [--quality 75] [--width 3840 --height 2160] [--keep-coefs]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import __graft_entry__ as ge, make_multiscan

ap = argparse.ArgumentParser()
ap.add_argument("--copies", type=int, default=1024)
ap.add_argument("--quality", type=int, default=75)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--keep-coefs", action="store_true")
a = ap.parse_args()
mjx = ge.load_package()
ctx = mjx.Context(0, profiling=True, throughput_plan=True)
src = mjx.synth_jpeg(a.width, a.height, "420", a.quality, seed=1)
forms = [("interleaved", src), ("three_scans", make_multiscan.twin(src)), ("luma_then_chroma", make_multiscan.twin(src, chroma_together=True))]
want = None
for name, data in forms:
    base = mjx.Batch(ctx, [mjx.ParsedScan(data)], keep_coefs=a.keep_coefs)
    assert all(s == mjx.OK for s in base.create_status), base.create_status
    batch = base.tile(a.copies) if a.copies > 1 else base
    batch.decode(); batch.wait()
    rgb = [batch.rgb(i) for i in (0, len(batch) - 1)]
    if want is None: want = rgb[0]
    same = all(np.array_equal(r, want) for r in rgb)          # the twins must give the source's picture bit for bit
    batch.kernel_ms(reset=True)
    unconv0 = batch.unconverged_runs()
    t0 = time.perf_counter()
    for _ in range(a.steps): batch.decode()
    batch.wait()
    el = time.perf_counter() - t0
    assert all(batch.status(i) == mjx.OK for i in range(len(batch)))
    unconv = batch.unconverged_runs() - unconv0            # (steps whose pictures were skipped: see bench.py)
    by, kms = batch.bytes(), batch.kernel_ms()
    print(json.dumps({"form": name, "file_bytes": len(data), "images": len(batch), "equal_to_interleaved": same, "unconverged_chunk_runs": unconv,
                      "Gpixels/s": round(by["pixels"] * a.steps / el / 1e9, 1), "ms_per_step": round(el / a.steps * 1e3, 3),
                      "geometry": base.geometry(),
                      "kernels_ms_per_step": {k: round(v[0] / a.steps, 3) for k, v in kms.items() if v[1]},
                      "launch_groups_per_step": {k: v[1] / a.steps for k, v in kms.items() if v[1]}}), flush=True)
    batch.close()
    if batch is not base: base.close()
