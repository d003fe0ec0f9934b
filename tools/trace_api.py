"""Host-side view of the last mjx_decode_batch call in a rocprofv3 --hip-trace --memory-copy-trace run of tools/e2e_from_files.py:
HIP API calls that took long, and when each big host-to-device copy was enqueued / started / finished.
python tools/trace_api.py DIR"""
import csv, glob, os, sys
d = sys.argv[1]
api = list(csv.DictReader(open(glob.glob(os.path.join(d, "*hip_api_trace.csv"))[0])))
mc = list(csv.DictReader(open(glob.glob(os.path.join(d, "*memory_copy_trace.csv"))[0])))
big = [r for r in mc if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 200000]
last = [big[-1]]
for r in reversed(big[:-1]):
    if int(last[0]["Start_Timestamp"]) - int(r["End_Timestamp"]) > 3e6:
        break
    last.insert(0, r)
t0 = int(last[0]["Start_Timestamp"]) - 1500000
t1 = int(last[-1]["End_Timestamp"]) + 4000000
ev = []
for r in last:
    ev.append((int(r["Start_Timestamp"]), "   DMA start (%.2f ms long)" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)))
    ev.append((int(r["End_Timestamp"]), "   DMA end"))
for r in api:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 or s > t1:
        continue
    n = r["Function"]
    if e - s > 40000 or n in ("hipMemcpyAsync",) and e - s > 5000 or n.startswith("hipMalloc") or n.startswith("hipFree") or n.startswith("hipHostMalloc"):
        ev.append((s, "%s (%.3f ms) tid %s" % (n, (e - s) / 1e6, r.get("Thread_Id", "?"))))
for t, what in sorted(ev):
    print("%8.3f  %s" % ((t - t0) / 1e6, what))
