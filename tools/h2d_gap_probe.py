"""How long the DMA engine idles between consecutive pinned host -> device copies on one stream, and whether two streams
overlap their copies (mjx_decode_batch enqueues one transfer per group of files: DESIGN.md s10)."""
import time, torch
total = 512 << 20
h = torch.empty(total, dtype=torch.uint8).pin_memory()
d = torch.empty(total, dtype=torch.uint8, device="cuda")
d.copy_(h, non_blocking=True); torch.cuda.synchronize()
def run(pieces, streams):
    ss = [torch.cuda.Stream() for _ in range(streams)]
    step = total // pieces
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(pieces):
        with torch.cuda.stream(ss[k % streams]):
            d[k * step:(k + 1) * step].copy_(h[k * step:(k + 1) * step], non_blocking=True)
    torch.cuda.synchronize()
    return time.perf_counter() - t
for streams in (1, 2, 4):
    for pieces in (1, 2, 4, 8, 16, 32, 64):
        best = min(run(pieces, streams) for _ in range(4))
        print("512 MB in %2d copies on %d stream(s): %.2f ms = %.1f GB/s" % (pieces, streams, best * 1e3, total / best / 1e9))
