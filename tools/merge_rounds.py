#!/usr/bin/env python3
"""Per-launch durations of the synchronisation kernels from a rocprofv3 kernel trace (one-stream run):
   tools/merge_rounds.py out_kernel_trace.csv   -> mean duration by position of the launch inside its chunk pass."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq, out, i = [], collections.defaultdict(list), 0
for r in rows:
    n = r["Kernel_Name"].split("(")[0]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if n.startswith("k_huff_spec"): i = 0
    if n.startswith("k_huff_"):
        out[(i, n)].append(d); i += 1
for (i, n), v in sorted(out.items()):
    print(f"{i:3d} {n:22s} n={len(v):3d} mean={sum(v)/len(v):9.1f} us  min={min(v):9.1f}")
