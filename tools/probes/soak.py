"""Round 5 probe: create / decode / free cycles of mixed batches through every front door for a minute; device memory before and after."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import __graft_entry__ as ge, make_multiscan
mjx = ge.load_package()
rng = np.random.default_rng(5)
free0 = torch.cuda.mem_get_info(0)[0]
ctx = mjx.Context(0)
datas = []
for i in range(48):
    w, h = int(rng.integers(16, 1400)), int(rng.integers(16, 1000))
    datas.append(mjx.synth_jpeg(w, h, ["420", "422", "444", "gray"][i % 4], int(rng.integers(30, 97)), seed=i))
datas += [make_multiscan.twin(datas[k], chroma_together=bool(k & 1)) for k in (0, 1, 4, 5)]
t0, n, px = time.time(), 0, 0
while time.time() - t0 < float(sys.argv[1]) if len(sys.argv) > 1 else 60.0:
    sel = [datas[int(j)] for j in rng.integers(0, len(datas), int(rng.integers(1, 64)))]
    mode = n % 3
    if mode == 0:
        b = mjx.Batch(ctx, [mjx.ParsedScan(d) for d in sel], keep_coefs=bool(n & 4), chunk_images=int(rng.integers(0, 40)))
        for _ in range(int(rng.integers(1, 4))): b.decode()
        b.wait()
        assert all(b.status(i) == mjx.OK for i in range(len(b)))
        b.close()
    elif mode == 1:
        b, st = mjx.decode_batch(ctx, sel, threads=4)
        assert all(s == mjx.OK for s in st)
        b.close()
    else:
        img = mjx.decode(sel[0])
    n += 1
ctx.close()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info(0)[0]
print("soak ok: %d rounds in %.0f s; free device memory before %.2f GB, after (context closed) %.2f GB" % (n, time.time() - t0, free0 / 1e9, free1 / 1e9))
