"""Round 5 probe: latency of small host-to-device copies (pinned and pageable) through torch, per process: on this pool the second
and later processes on a box see 20-30 ms for copies of 128 KB and more where the first sees 0.1 ms."""
import sys, time, torch
dev = torch.device("cuda:0")
torch.zeros(1, device=dev); torch.cuda.synchronize()
for pinned in (True, False):
    for n in (32 << 10, 100 << 10, 128 << 10, 160 << 10, 256 << 10, 1 << 20, 8 << 20):
        h = torch.empty(n, dtype=torch.uint8, pin_memory=pinned)
        d = torch.empty(n, dtype=torch.uint8, device=dev)
        ts = []
        for _ in range(8):
            t = time.perf_counter(); d.copy_(h, non_blocking=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        print("pinned" if pinned else "pageable", "%8d B" % n, " ".join("%.2f" % (x * 1e3) for x in ts[2:]), "ms")
