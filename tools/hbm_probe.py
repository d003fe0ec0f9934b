"""Practical HBM ceilings of the box (torch fill / copy / read-reduce on 16 GiB buffers): python tools/hbm_probe.py"""
import torch, time
n = 16 * 1024**3
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
dt = t(lambda: a.fill_(7)); print("fill  : %.2f TB/s written" % (n / dt / 1e12))
dt = t(lambda: b.copy_(a)); print("copy  : %.2f TB/s read + %.2f TB/s written = %.2f" % (n / dt / 1e12, n / dt / 1e12, 2 * n / dt / 1e12))
v = a.view(torch.int32)
dt = t(lambda: v.sum()); print("sum   : %.2f TB/s read" % (n / dt / 1e12))
