"""Round 5 probe: a dumped fuzz batch decoded with and without kept coefficients (multi-scan pictures: gathered / read from their
scans' streams); prints where the pictures differ.   python tools/probes/planar_batch.py DIR [first last]"""
import os, re, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
mjx = ge.load_package()
d = sys.argv[1]
head = open(os.path.join(d, "batch.txt")).readline()
chunk = int(re.search(r"chunk_images (\d+)", head).group(1))
files = sorted(f for f in os.listdir(d) if f.endswith(".jpg"))
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, len(files))
files = files[lo:hi]
datas = [open(os.path.join(d, f), "rb").read() for f in files]
ctx = mjx.Context(0)
scans = [mjx.ParsedScan(x) for x in datas]
a = mjx.Batch(ctx, scans, keep_coefs=True, chunk_images=chunk); a.decode(); a.wait()
b = mjx.Batch(ctx, scans, chunk_images=chunk); b.decode(); b.wait()
print("files", len(files), "geometry", b.geometry())
for i, f in enumerate(files):
    if a.status(i) != b.status(i):
        print(f, "status", a.status(i), b.status(i)); continue
    if a.status(i): continue
    x, y = a.rgb(i).astype(int), b.rgb(i).astype(int)
    if not np.array_equal(x, y):
        ys, xs = np.nonzero(np.abs(x - y).max(axis=2) > 0)
        m = sorted(set(zip((ys // 16).tolist(), (xs // 16).tolist())))
        print(f, x.shape, "differs in", len(ys), "pixels; MCUs (row, col):", m[:24], "...", len(m), "n_parts", scans[i].desc.n_parts)
print("done")
