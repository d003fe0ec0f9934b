#!/bin/bash
# FETCH_SIZE of two builds on one box: tools/probes/fetch_ab.sh outdir lib1 lib2
O=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for L in "$@"; do
  n=$(basename $L .so)
  export MJX_LIB=$PWD/$L
  rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_$n -o out --output-format csv -- python3 bench.py --no-cpu-baseline --no-extra --no-parity --steps 1 --warmup 0 --images-per-gpu 256 > $O/pmc_fetch_$n.log 2>&1
  python3 - $O/pmc_fetch_$n $n <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            k = r["Kernel_Name"].split("(")[0].replace("void mjx::", "").split("<")[0]
            acc[k][0] += float(r["Counter_Value"]) * 1024; acc[k][1] += 1
print(sys.argv[2], {k: "%.3f GB x2 / %d launches" % (v / n / 1e9 * 2, n) for k, (v, n) in sorted(acc.items()) if k.startswith("k_")})
PY
done
