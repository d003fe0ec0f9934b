cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for L in "$@"; do
MJX_LIB=$PWD/ab/libmjx_$L.so rocprofv3 --kernel-trace -d gpurun_out/prof_$L -o out --output-format csv -- python3 bench.py --no-cpu-baseline --no-extra --no-traffic --steps 1 --warmup 1 > gpurun_out/prof_$L.log 2>&1
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/prof_$L/out_kernel_trace.csv")) if "huff" in r["Kernel_Name"]]
print("$L", [(r["Kernel_Name"][7:17], round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)) for r in rows[-11:]])
PY
done
