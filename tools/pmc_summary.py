"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (sum of counter values over dispatches)."""
import csv, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for path in sys.argv[1:]:
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (path, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); calls[(path, k)] += 1
for k, c in rows.items():
    print(k, {n: int(v) for n, v in c.items()})
    d = c
    if "SQ_INSTS_VALU" in d and d.get("SQ_ACTIVE_INST_VALU"):
        print("   lane utilisation (THREAD_CYCLES_VALU / ACTIVE_INST_VALU / 64): %.3f" % (d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"] / 64))
        print("   VALU busy share of wave cycles: %.3f ; wait_any %.3f ; wait_inst_any %.3f ; valu insts/wave %.0f" % (
            d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"], d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], d["SQ_INSTS_VALU"] / d["SQ_WAVES"]))
        print("   avg waves in flight per SIMD (WAVE_CYCLES / BUSY_CYCLES-ish): %.2f" % (d["SQ_WAVE_CYCLES"] / d["SQ_BUSY_CYCLES"]))
    if "SQ_INSTS_LDS" in d and d.get("SQ_INSTS_LDS"):
        print("   LDS: insts %d, bank conflict cycles / idx active %.3f, wait_inst_lds share of active_any %.3f" % (
            d["SQ_INSTS_LDS"], d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1), d["SQ_WAIT_INST_LDS"] / max(d["SQ_ACTIVE_INST_ANY"], 1)))
