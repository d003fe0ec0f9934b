#!/bin/bash
# Counters behind the write pass's store cost (run through gpurun): tools/probes/pmc_write_stores.sh OUTDIR lib1 lib2 ...
O=$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MJX_STREAMS=1 MJX_BENCH_IGNORE_STATUS=1
Q="--no-cpu-baseline --no-extra --no-parity --steps 1 --warmup 0 --images-per-gpu 256"
for L in "$@"; do
  n=$(basename $L .so)
  export MJX_LIB=$PWD/$L
  i=0
  for set in "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL" \
             "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TAG_STALL" \
             "TCP_TCC_WRITE_REQ TCP_TCC_WRITE_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_TOTAL_WRITE" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
    i=$((i+1))
    rocprofv3 --pmc $set -d $O/p_${n}_$i -o out --output-format csv -- python3 bench.py $Q > $O/p_${n}_$i.log 2>&1
  done
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections, os
O = sys.argv[1]
acc = collections.defaultdict(dict)
for d in sorted(glob.glob(O + "/p_*_*/")):
    lib = os.path.basename(d.rstrip("/")).rsplit("_", 1)[0][2:]
    for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "huff_write" not in k and "idct_color" not in k and "huff_spec" not in k:
                continue
            key = (lib, k.replace("void mjx::", "").split("<")[0])
            acc[key][r["Counter_Name"]] = acc[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
with open(O + "/summary.txt", "w") as out:
    for key in sorted(acc):
        out.write("%s %s\n" % key)
        for c, v in sorted(acc[key].items()):
            out.write("    %-36s %.4g\n" % (c, v))
print(open(O + "/summary.txt").read())
PY
rm -rf $O/p_*_*/
