"""Raw pinned host -> device bandwidth of this box (ceiling for mjx_decode_batch from host bytes)."""
import time, torch
for mb in (16, 96, 512):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print("H2D %4d MB pinned: %.2f ms = %.1f GB/s" % (mb, dt * 1e3, (mb << 20) / dt / 1e9))
