# Round 5 probe: synchronisation rounds enqueued up front on the headline (bench.py refuses a line with unconverged runs)
for r in 1 2; do for p in 6 5 4 3; do echo -n "rounds=$p "; MJX_FIX_PASSES=$p python bench.py --no-cpu-baseline --no-extra --no-parity 2>&1 | grep '^{\|Assert' | python3 -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(round(d['value']), d['ms_per_step'], d['unconverged_chunk_runs'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})
except Exception: print(t[15:80])"; done; done
