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
        # SQ_BUSY_CYCLES is summed over the chip's 32 shader engines.  One price for a wave64 vector instruction since round 6
        # (tools/probes/op_cost_probe.hip): 4 cycles of its SIMD -- what the packed fp32, conversion, shift-left, bit-field,
        # 24-bit multiply and compare instructions these kernels are made of take whatever the occupancy; the 2-cycle figure (plain
        # f32 add / mul / fma, mov, and, add with two or more waves ready) is printed beside it as the lower bound.
        cyc = d["SQ_BUSY_CYCLES"] / 32.0
        print("   vector issue share of SIMD time at 4 cycles per instruction: %.3f   (at 2 cycles: %.3f)" % (
            d["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cyc), d["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cyc)))
    if d.get("SQ_INSTS_SALU") and d.get("SQ_INSTS_VALU"):
        print("   scalar instructions per vector instruction: %.2f" % (d["SQ_INSTS_SALU"] / d["SQ_INSTS_VALU"]))
    if "SQ_INSTS_LDS" in d and d.get("SQ_INSTS_LDS"):
        print("   LDS: insts %d, bank conflict cycles / idx active %.3f, wait_inst_lds share of active_any %.3f" % (
            d["SQ_INSTS_LDS"], d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1), d["SQ_WAIT_INST_LDS"] / max(d["SQ_ACTIVE_INST_ANY"], 1)))
