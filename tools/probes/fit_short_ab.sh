# Round 5 probe: scans below one workgroup cut shorter (MJX_FIT_SHORT, mjx_plan.cpp) against the default cut, with more
# synchronisation rounds enqueued up front; bench.py refuses a timed region in which a chunk had not converged.
for f in "1 6" "1 10" "1 16" "0 6"; do set -- $f
for wh in "512 512 16384" "1024 768 8192"; do set -- $f $wh; echo "fit=$1 rounds=$2 $3x$4"; MJX_FIX_PASSES=$2 MJX_FIT_SHORT=$1 python bench.py --no-cpu-baseline --no-extra --no-parity --width $3 --height $4 --images-per-gpu $5 2>&1 | grep '^{\|Assert' | python3 -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(round(d['value']), d['ms_per_step'], d['unconverged_chunk_runs'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})
except Exception: print(t[-200:])"; done
done
