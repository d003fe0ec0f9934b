"""Reads a rocprofv3 --kernel-trace csv of a bench run and prints, for the timed steps, how the kernels overlap: wall time covered
by at least one kernel, by two or more kernel classes at once, and the time every class is running.
    rocprofv3 --kernel-trace -d DIR -o out --output-format csv -- python3 bench.py --no-cpu-baseline --no-extra --no-parity --steps 4
    python tools/trace_overlap.py DIR/*kernel_trace.csv"""
import csv, sys, collections
rows = []
for f in [a for a in sys.argv[1:] if not a.isdigit()]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void mjx::", "").split("<")[0]
        if not k.startswith("k_"):
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
# the timed steps: the last dense cluster of k_idct_color launches; take the last 60 % of the span of all k_huff_spec launches
spec = [r for r in rows if r[2] in ("k_huff_spec", "k_huff_emit")]      # (the first kernel of a chunk's entropy stage: two passes / single decode)
t0 = spec[len(spec) // 2][0]
t1 = max(r[1] for r in rows if r[2] == "k_idct_color")
sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
ev = []
for a, b, k in sel:
    ev.append((a, 1, k)); ev.append((b, -1, k))
ev.sort()
active = collections.Counter()
last = t0
cover = collections.Counter()   # number of distinct classes -> ns
busy = collections.Counter()
group = lambda k: "pixels" if k in ("k_idct_color", "k_ref_color") else "entropy"
pair = collections.Counter()
for t, d, k in ev:
    dt = t - last
    if dt > 0:
        classes = [c for c, n in active.items() if n > 0]
        cover[len(classes)] += dt
        for c in classes:
            busy[c] += dt
        g = {group(c) for c in classes}
        pair["+".join(sorted(g)) or "idle"] += dt
    active[k] += d
    last = t
span = t1 - t0
print("span %.2f ms" % (span / 1e6))
for n in sorted(cover):
    print("  %d kernel classes running: %5.1f %%" % (n, 100.0 * cover[n] / span))
for g, v in sorted(pair.items()):
    print("  %-16s %5.1f %%" % (g, 100.0 * v / span))
for c, v in sorted(busy.items(), key=lambda x: -x[1]):
    print("  %-20s running %5.1f %% of the span" % (c, 100.0 * v / span))

# the long launches at the end of the span, on the span's clock: where the stages really lie beside each other
# (a trailing number on the command line = how many to print)
count = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 14
longk = [r for r in sel if r[1] - r[0] > 500000]
print("launches longer than 0.5 ms (start .. end in ms from the span's start):")
for a, b, k in longk[-count:]:
    print("  %-18s %8.2f .. %8.2f  (%.2f ms)" % (k, (a - t0) / 1e6, (b - t0) / 1e6, (b - a) / 1e6))
