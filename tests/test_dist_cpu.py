"""CPU test of the N>1 path (gloo, world_size 2): the batch shards by image with no data-path collective; the only
collective is the max-reduce of the timed region.  Checks that the ranks' shards partition the global batch."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, unique, per_gpu, q):
    import torch
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seeds = bench.shard_seeds(rank, world, unique)
    reps = per_gpu // len(seeds)
    mine = [(rank + world * j, seeds[j % len(seeds)]) for j in range(reps * len(seeds))]   # (global index, seed)
    # every global image must carry seed i % unique
    assert all(s == g % unique for g, s in mine)
    cnt = torch.tensor([len(mine), sum(g for g, _ in mine)], dtype=torch.int64)
    dist.all_reduce(cnt)                                   # test-only reduction to check the partition
    elapsed = bench.reduce_elapsed(1.0 + rank, world)      # the job's only collective
    q.put((rank, int(cnt[0]), int(cnt[1]), elapsed, len(seeds)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("unique,per_gpu", [(64, 2048), (8, 64)])
def test_two_rank_sharding_partitions_the_batch(unique, per_gpu):
    import torch.multiprocessing as mp
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, unique, per_gpu, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    total = world * per_gpu
    for rank, n, idx_sum, elapsed, period in res:
        assert n == total and idx_sum == total * (total - 1) // 2      # shards are disjoint and cover 0..total-1
        assert elapsed == 2.0                                          # MAX over ranks
        assert period == unique // 2


def test_shard_seeds_single_gpu():
    import bench
    assert bench.shard_seeds(0, 1, 64) == list(range(64))
    assert bench.shard_seeds(3, 8, 64) == [3 + 8 * j for j in range(8)]
