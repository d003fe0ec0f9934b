cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for L in "$@"; do
rm -rf gpurun_out/pmct_$L
MJX_LIB=$PWD/ab/libmjx_$L.so rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD -d gpurun_out/pmct_$L -o out --output-format csv -- python3 bench.py --no-cpu-baseline --no-extra --no-traffic --steps 1 --warmup 0 --images-per-gpu 1024 > gpurun_out/pmct_$L.log 2>&1
python3 - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
first={}
for r in csv.DictReader(open("gpurun_out/pmct_$L/out_counter_collection.csv")):
    k=r["Kernel_Name"].split("(")[0]
    if "merge" not in k: continue
    d=int(r["Dispatch_Id"])
    first.setdefault(k, d)
    if d==first[k]: acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items(): print("$L", k, {n:int(x) for n,x in v.items()})
PY
done
