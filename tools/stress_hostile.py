"""Hostile-input stress run for the device path (GPU box only): python tools/stress_hostile.py [seed]

Mutates the scan data of the restart-interval, multi-scan and plain fixtures (byte flips that favour 0xFF / RSTn, truncation,
64-byte random overwrites, deletions), decodes everything in randomly sized chunks and only requires that the
process survives and every image reports a status.  tests/test_gpu_parity.py holds the bounded version of this.
"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
mjx = ge.load_package()
ctx = mjx.Context(0)
d = os.path.join(ROOT, "tests", "golden", "pil")
names = ["dri_420_r5", "dri_420_720p_rows", "dri_444_r1", "dri_422_rows", "dri_gray_r7", "dri_420_r300", "opt_420_q85", "std_420_big",
         "ms_420_big", "ms_420_q85_rst", "ms_422_q95", "ms2_420_big", "ms2_420_q85_rst"]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
tot = 0
for rnd in range(12):
    scans = []
    for name in names:
        base = open(os.path.join(d, name + ".jpg"), "rb").read()
        sos = base.index(b"\xff\xda")
        for k in range(16):
            b = bytearray(base)
            mode = rng.integers(0, 4)
            if mode == 0:
                for _ in range(int(rng.integers(1, 12))):
                    b[int(rng.integers(sos + 14, len(b)))] = int(rng.choice([0xff, 0xd0, 0xd3, 0x00, int(rng.integers(0, 256))]))
            elif mode == 1:
                b = b[: int(rng.integers(sos + 20, len(b)))]
            elif mode == 2:
                i = int(rng.integers(sos + 14, len(b) - 64)); b[i:i + 64] = bytes(rng.integers(0, 256, 64, dtype=np.uint8))
            else:
                i = int(rng.integers(sos + 14, len(b) - 8)); del b[i:i + int(rng.integers(1, 200))]
            try:
                scans.append(mjx.ParsedScan(bytes(b)))
            except mjx.MjxError:
                pass
    batch = mjx.Batch(ctx, scans, chunk_images=int(rng.integers(3, 40)))
    batch.decode(); batch.wait()
    st = [batch.status(i) for i in range(len(scans))]
    tot += len(scans)
    batch.close()
print("stress ok", tot)
