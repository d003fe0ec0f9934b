"""Round 5 probe: a device-to-host copy into pageable memory followed by small work and a stream synchronisation, per process."""
import sys, time, torch
dev = torch.device("cuda:0")
x = torch.zeros(1, device=dev); torch.cuda.synchronize()
s2 = torch.cuda.Stream()
for n in (786432, 1338750, 4 << 20):
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    for pinned in (False, True):
        h = torch.empty(n, dtype=torch.uint8, pin_memory=pinned)
        ts = []
        for _ in range(8):
            t = time.perf_counter()
            h.copy_(d)                                   # synchronous D2H
            t1 = time.perf_counter()
            with torch.cuda.stream(s2):
                x.zero_()                                # a fill on another stream
            s2.synchronize()
            ts.append((t1 - t, time.perf_counter() - t1))
        print("D2H %8d B to %s: copy %s ms | fill + sync on another stream %s ms" % (n, "pinned  " if pinned else "pageable",
              " ".join("%.2f" % (a * 1e3) for a, _ in ts[2:]), " ".join("%.2f" % (b * 1e3) for _, b in ts[2:])))
