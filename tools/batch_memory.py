"""Device memory a 2048 x 4K batch holds (run on a GPU box): python tools/batch_memory.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
mjx = ge.load_package()
ctx = mjx.Context(0, throughput_plan=True)        # (the 64 originals are cut like the batch they are tiled into, as bench.py does)
free0, total = torch.cuda.mem_get_info()
datas = mjx.synth_batch(64, 3840, 2160, "420", 75)
base = mjx.Batch(ctx, [mjx.ParsedScan(d) for d in datas])
big = base.tile(32)
big.decode(); big.wait()
free1, _ = torch.cuda.mem_get_info()
print("device memory in use by the 2048 x 4K batch: %.1f GB (of %.0f)" % ((free0 - free1) / 1e9, total / 1e9), big.geometry(), big.bytes())
