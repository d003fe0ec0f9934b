"""Timeline summary of a rocprofv3 trace of tools/e2e_from_files.py (kernel + memory-copy CSVs in one directory):
python tools/trace_e2e.py DIR  -> the big host-to-device copies of the last mjx_decode_batch call, busy time per kernel and per stream."""
import csv, glob, os, sys, collections
d = sys.argv[1]
mc = list(csv.DictReader(open(glob.glob(os.path.join(d, "*memory_copy_trace.csv"))[0])))
kt = list(csv.DictReader(open(glob.glob(os.path.join(d, "*kernel_trace.csv"))[0])))
big = [r for r in mc if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 200000]
# the last call: walk back from the last big copy while the gap to the previous one stays below 5 ms
last = [big[-1]]
for r in reversed(big[:-1]):
    if int(last[0]["Start_Timestamp"]) - int(r["End_Timestamp"]) > 5e6:
        break
    last.insert(0, r)
base = int(last[0]["Start_Timestamp"])
busy = 0.0
for r in last:
    s = (int(r["Start_Timestamp"]) - base) / 1e6
    e = (int(r["End_Timestamp"]) - base) / 1e6
    busy += e - s
    print("H2D stream %s  %.2f -> %.2f (%.2f ms)" % (r["Stream_Id"], s, e, e - s))
ks = [r for r in kt if int(r["Start_Timestamp"]) >= base]
end = max(int(r["End_Timestamp"]) for r in ks)
print("window %.2f ms, %d kernels, copies busy %.2f ms" % ((end - base) / 1e6, len(ks), busy))
per = collections.defaultdict(float)
per_stream = collections.defaultdict(float)
for r in ks:
    dt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    per[r["Kernel_Name"][:32]] += dt
    per_stream[r["Stream_Id"]] += dt
for k, v in sorted(per.items(), key=lambda x: -x[1])[:12]:
    print("%-34s %.2f ms" % (k, v))
for k, v in sorted(per_stream.items()):
    last_end = max(int(r["End_Timestamp"]) for r in ks if r["Stream_Id"] == k)
    print("stream %s: busy %.2f ms, last kernel ends at %.2f" % (k, v, (last_end - base) / 1e6))
