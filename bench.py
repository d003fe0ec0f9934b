#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the baseline-JPEG decode hot path on 4K 4:2:0 batches (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A *step* = one pass of the whole hot path (entropy decode -> dequant -> IDCT -> upsample -> RGB) over the rank's batch:
`--images-per-gpu` synthetic 3840x2160 4:2:0 q75 baseline JPEGs (BASELINE.json configs[4] is 16384 images over 8 GPUs
= 2048 per GPU; per-GPU work is fixed as N grows -> weak scaling).  Inputs (de-stuffed scans, tables) are resident in
HBM before the timed region; outputs stay in HBM.  Images are independent, so ranks never exchange data: image i of the
global batch goes to rank i mod N; the only collective is the reporting barrier / max-reduce of the elapsed time.

The JSON line carries, besides the contract fields:
  roofline      dominant kernel (by summed HIP-event time over the timed region): algorithmic bytes per launch / average
                launch duration vs the 8 TB/s HBM3E peak
  roofline_e2e  whole path: sum over images of (S + 3*W*H) (SURVEY.md s8(d) B_e2e) / sum of all kernel time
  kernels       per kernel class: launches, total ms
  cpu_baseline  the oracle (C restatement of the reference's algorithm, kind "port") on the host cores, rank 0, N=1
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images-per-gpu", type=int, default=2048)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--subsampling", default="420")
    ap.add_argument("--quality", type=int, default=75)
    ap.add_argument("--unique", type=int, default=64, help="distinct synthetic images in the global batch (SURVEY s8(d))")
    ap.add_argument("--chunk-images", type=int, default=0)
    ap.add_argument("--stages", default="all", choices=["all", "pixels"], help="pixels = stage-B-only sweep on resident coefficients")
    ap.add_argument("--device-destuff", action="store_true", help="upload stuffed scans; FF00 compaction on the GPU at upload")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2], help="2 = odd chunks on a second HIP stream (stage B of one chunk overlaps stage A of the next; per-kernel times then include the contention)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    return ap.parse_args()


def algorithmic_bytes(kind, by, nsub_total, nblk):
    """Per-step algorithmic bytes of each kernel class; by = mjx.Batch.bytes().
    by["coef"] = bytes of the intermediate coefficient representation (4 per stream entry + 4 per block of DC).
    idct_color uses SURVEY.md s8(d)'s stage-B figure B_idct = 128*n_blocks + 3*W*H (6 B/px for 4:2:0); the kernel's
    real input is the smaller compact stream -- `algorithmic_bytes_physical` gives that variant."""
    S, rgb, coef = by["scan"], by["rgb"], by["coef"]
    state = 32 * nsub_total                       # entry + exit state per subsequence
    cps = 8 * 15 * nsub_total                     # two checkpoint words every 256 bits
    return {
        "gather": 2 * coef,                       # multi-scan pictures only: component streams in, picture stream out
        "huff_sync": S + state + cps,             # k_huff_spec: scan in, states + checkpoints out
        "huff_fix": S // 5 + state + cps // 5,    # k_huff_merge rounds: ~1/5 of the scan is re-read (median merge distance)
        "huff_scan": 24 * nsub_total,             # exit state in, block base + entry base out
        "huff_write": S + 24 * nsub_total + coef, # scan + entry state + bases in, compact stream + DC out
        "dc_scan": 8 * nblk,                      # DC differences in, predicted DC out
        "idct_color": 128 * nblk + rgb,           # B_idct
    }[kind]


def algorithmic_bytes_physical(kind, by, nsub_total, nblk):
    if kind == "idct_color":
        return by["coef"] + by["rgb"]             # compact stream + DC in, packed RGB out
    return algorithmic_bytes(kind, by, nsub_total, nblk)


def measured_traffic(kind, images_per_launch):
    """HBM bytes per launch from the committed rocprofv3 PMC collection (profiles/*_traffic.json, FETCH_SIZE x 2 +
    WRITE_SIZE, separate passes), scaled to this run's images per launch.  None if no collection is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None
    t = json.load(open(files[-1]))
    k = t["kernels"].get(kind)
    if not k:
        return None
    return int(k["hbm_bytes"] * images_per_launch / t["images_per_launch"]), os.path.basename(files[-1])


def shard_seeds(rank, world, unique):
    """Content seeds of this rank's images: global image i goes to rank i % world and has seed i % unique, so the
    rank sees the periodic sequence (rank + world*j) % unique; returns one period."""
    seeds, j = [], 0
    while True:
        s = (rank + world * j) % unique
        if j > 0 and s == seeds[0]:
            break
        seeds.append(s)
        j += 1
    return seeds


def reduce_elapsed(elapsed, world, device=None):
    """MAX over ranks of the timed region (the only collective of the job; the data path has none)."""
    if world <= 1:
        return elapsed
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def cpu_baseline(mjx, datas, width, height, threads):
    import oracle_binding as orc
    cores = os.cpu_count() or 1
    threads = threads or min(cores, 32)
    sample = [datas[i % len(datas)] for i in range(threads)]
    t = time.perf_counter()
    px, st = orc.decode_many(sample, threads, layout=orc.LAYOUT_REF, faithful=True)
    dt = time.perf_counter() - t
    ok = sum(1 for s in st if s == 0)
    return {
        "value": round(px / dt / 1e6, 4), "unit": "Mpixels/s", "cores": threads, "kind": "port",
        "sample": "%d images of the same %dx%d batch, one per thread, reference algorithm restated in C "
                  "(oracle/: O(n^4) float IDCT with cosf per term, linear-search Huffman; gcc -O2 -ffp-contract=off); "
                  "%d decoded ok in %.1f s; host has %d cores" % (len(sample), width, height, ok, dt, cores),
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    ge.build()
    mjx = ge.load_package()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decode path has no CPU fallback")
    # one rank per GPU; MJX_BENCH_BACKEND=gloo + a single visible GPU is only for exercising the N>1 code path in tests
    backend = os.environ.get("MJX_BENCH_BACKEND", "nccl")
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    # ---- this rank's shard of the global batch: global image i -> rank i % world, content seed i % unique ----
    per_gpu = args.images_per_gpu
    seeds = shard_seeds(rank, world, args.unique)
    period = len(seeds)
    reps = max(1, per_gpu // period)
    per_gpu = reps * period
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
        datas = list(ex.map(lambda s: mjx.synth_jpeg(args.width, args.height, args.subsampling, args.quality, s), seeds))

    if args.streams == 2:
        os.environ["MJX_STREAMS"] = "2"
    ctx = mjx.Context(device, profiling=True)
    # host side of the boundary (not part of `value`): marker walk + de-stuffing of the unique files on the
    # host cores, then planning + table construction + upload of the compressed scans (mjx_batch_create)
    t_h = time.perf_counter()
    with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
        scans = list(ex.map(lambda d: mjx.ParsedScan(d, device_destuff=args.device_destuff), datas))
    t_parse = time.perf_counter() - t_h
    keep = args.stages == "pixels"
    t_h = time.perf_counter()
    base = mjx.Batch(ctx, scans, keep_coefs=keep, chunk_images=args.chunk_images)
    t_create = time.perf_counter() - t_h
    assert all(s == mjx.OK for s in base.create_status), base.create_status
    host_side = {"files": len(datas), "parse_ms_per_file": round(1e3 * t_parse / len(datas), 3),
                 "parse_threads": min(32, os.cpu_count() or 1),
                 "create_ms_per_file": round(1e3 * t_create / len(datas), 3),
                 "compressed_MB_per_file": round(sum(len(d) for d in datas) / len(datas) / 1e6, 3),
                 "note": "host marker walk + de-stuffing (threads), then planning + decode tables + H2D of the compressed scans "
                         "and first-use allocations (mjx_batch_create, one thread) for the unique files; not part of `value`"}
    batch = base.tile(reps) if reps > 1 else base
    if batch is not base:
        base.close()
    nsub_total = sum((len(mjx.ParsedScan(d).scan_bytes()) + 511) // 512 for d in datas) * reps
    nblk = sum(batch.info(i)["bpm"] * batch.info(i)["mcus"] for i in range(period)) * reps
    stages = mjx.STAGE_ALL if args.stages == "all" else mjx.STAGE_PIXELS
    if args.stages == "pixels":
        batch.decode(mjx.STAGE_ALL)
        batch.wait()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.decode(stages)
        batch.wait()
    batch.kernel_ms(reset=True)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.decode(stages)
    batch.wait()
    sync_all()
    elapsed = time.perf_counter() - t0
    elapsed = reduce_elapsed(elapsed, world, "cuda" if backend == "nccl" else None)
    bad = [i for i in range(len(batch)) if batch.status(i) != mjx.OK]
    assert not bad, "images failed: %s" % bad[:8]

    kms = batch.kernel_ms()
    by = batch.bytes()
    total_px = by["pixels"] * world * args.steps
    value = total_px / elapsed / 1e6
    kernels = {k: {"launches": v[1], "ms": round(v[0], 4)} for k, v in kms.items() if v[1]}
    out = {
        "metric": "Mpixels/sec decode, 4K 4:2:0 baseline batch", "value": round(value, 2), "unit": "Mpixels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%d x %dx%d %s baseline JPEG q%d per GPU (%d unique, tiled on device), de-stuffed scans "
                               "resident in HBM, RGB out in HBM, stages=%s"
                               % (per_gpu, args.width, args.height,
                                  {"420": "4:2:0", "422": "4:2:2", "444": "4:4:4", "440": "4:4:0", "gray": "greyscale"}[args.subsampling],
                                  args.quality, period, args.stages),
                   "images_per_gpu": per_gpu, "width": args.width, "height": args.height, "subsampling": args.subsampling,
                   "quality": args.quality, "layout": "standard", "streams": args.streams, "sharding": "image i -> gpu i %% %d, no collective" % world},
        "kernels": kernels,
    }
    if kernels:
        dom = max(kernels, key=lambda k: kernels[k]["ms"])
        n_launch = kernels[dom]["launches"]
        avg_s = kernels[dom]["ms"] / 1e3 / n_launch
        per_launch = algorithmic_bytes(dom, by, nsub_total, nblk) * args.steps / n_launch
        per_launch_phys = algorithmic_bytes_physical(dom, by, nsub_total, nblk) * args.steps / n_launch
        ach = per_launch / avg_s / 1e9
        imgs_per_launch = per_gpu * args.steps / n_launch * (4 if dom == "huff_fix" else 1)
        tr = measured_traffic(dom, imgs_per_launch) if args.width == 3840 and args.height == 2160 and args.quality == 75 else None
        out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": tr[0] if tr else None,
                           "traffic_source": tr[1] if tr else None,
                           "bytes_per_launch": int(per_launch), "avg_launch_ms": round(avg_s * 1e3, 5),
                           "achieved_physical": round(per_launch_phys / avg_s / 1e9, 2),
                           "bytes_basis": "SURVEY s8(d) B_idct = 128*n_blocks + 3*W*H" if dom == "idct_color" else "DESIGN.md s5"}
        # every kernel class, same definitions
        out["kernel_rooflines"] = {
            k: {"GB/s": round(algorithmic_bytes(k, by, nsub_total, nblk) * args.steps / (v["ms"] / 1e3) / 1e9, 1),
                "frac": round(algorithmic_bytes(k, by, nsub_total, nblk) * args.steps / (v["ms"] / 1e3) / 1e9 / HBM_PEAK_GBS, 4)}
            for k, v in kernels.items() if v["ms"] > 0}
        tot_ms = sum(v["ms"] for v in kernels.values())
        e2e_bytes = (by["scan"] + by["rgb"]) if args.stages == "all" else algorithmic_bytes("idct_color", by, nsub_total, nblk)
        e2e = e2e_bytes * args.steps / (tot_ms / 1e3) / 1e9
        out["roofline_e2e"] = {"bound": "hbm", "achieved": round(e2e, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(e2e / HBM_PEAK_GBS, 5), "bytes_per_step": int(e2e_bytes),
                               "kernel_ms_per_step": round(tot_ms / args.steps, 4),
                               "definition": "sum(S + 3*W*H) / sum of kernel time" if args.stages == "all" else "B_idct = 128*n_blocks + 3*W*H (SURVEY s8(d)) / kernel time"}
    out["host_side"] = host_side
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(mjx, datas, args.width, args.height, args.cpu_threads)
    batch.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
