#!/bin/bash
# (TA_* counters abort rocprofv3 on this pool and hang its shutdown: left out; every pass under its own timeout)
# LDS / cache counters of stage B for several builds (run through gpurun): tools/probes/pmc_stage_b.sh OUTDIR lib1 lib2 ...
O=$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MJX_STREAMS=1 MJX_BENCH_IGNORE_STATUS=1
DEFAULT_SETS="SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_WAVE_CYCLES
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum
TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
# (PMC_SETS: newline-separated counter sets instead of these; MJX_STREAM_LINEAR etc. are inherited by the runs)
Q="--no-cpu-baseline --no-extra --no-parity --steps 1 --warmup 0 --images-per-gpu ${PMC_IMAGES:-256}"
for L in "$@"; do
  n=$(basename $L .so)
  export MJX_LIB=$PWD/$L
  i=0
  while read -r set; do
    [ -z "$set" ] && continue
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $set -d $O/p_${n}_$i -o out --output-format csv -- python3 bench.py $Q > $O/p_${n}_$i.log 2>&1 < /dev/null
  done <<< "${PMC_SETS:-$DEFAULT_SETS}"
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
            if "idct_color" not in k and "huff_write" not in k:
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
