# Round 5 probe: how many synchronisation rounds must be enqueued up front for batches of small pictures / multi-scan files
for r in 6 7 8 9 10; do
for wh in "256 256 32768" "512 512 16384" "1024 768 8192" "1280 720 8192"; do set -- $wh; echo -n "rounds=$r $1x$2 "; MJX_FIX_PASSES=$r python bench.py --no-cpu-baseline --no-extra --no-parity --width $1 --height $2 --images-per-gpu $3 2>&1 | grep '^{\|Assert' | python3 -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(round(d['value']), d['ms_per_step'], d['config']['subsequence_bytes'])
except Exception: print(t[15:60])"; done
echo -n "rounds=$r multiscan "; MJX_FIX_PASSES=$r python tools/bench_multiscan.py 2>&1 | tail -2 | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['form'], d['Gpixels/s'], d['unconverged_chunk_runs'], end=' | ')
print()"
done
