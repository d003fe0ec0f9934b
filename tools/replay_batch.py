"""Replays a batch dumped by tools/fuzz_parity.py (its third argument) and checks it against the oracle:
    python tools/replay_batch.py DIR          (stuffed / chunk_images are read from DIR/batch.txt)"""
import os, re, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import oracle_binding as orc
mjx = ge.load_package()
d = sys.argv[1]
head = open(os.path.join(d, "batch.txt")).readline()
stuffed = bool(int(re.search(r"stuffed (\d)", head).group(1))) if "stuffed" in head else False
chunk = int(re.search(r"chunk_images (\d+)", head).group(1))
files = sorted(f for f in os.listdir(d) if f.endswith(".jpg"))
datas = [open(os.path.join(d, f), "rb").read() for f in files]
ctx = mjx.Context(0)
scans = [mjx.ParsedScan(x, device_destuff=stuffed) for x in datas]
b = mjx.Batch(ctx, scans, keep_coefs=True, chunk_images=chunk)
b.decode(); b.wait()
bad = 0
for i, x in enumerate(datas):
    try:
        ref = orc.decode(x, layout=orc.LAYOUT_STD, ext_dri=True, ext_1bit=True, ext_multiscan=True)
    except orc.OracleError as e:
        print(files[i], "oracle refuses:", e, "device status", b.status(i)); continue
    st = b.status(i)
    t0 = st == 0 and np.array_equal(b.coefs(i), orc.interleave(ref))
    if not t0:
        bad += 1
        print(files[i], "status", st, "T0", t0, "scan bytes", scans[i].desc.scan_len, ref.rgb.shape)
print("stuffed", stuffed, "chunk_images", chunk, "files", len(files), "bad", bad)
