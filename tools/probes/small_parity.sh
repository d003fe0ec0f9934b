# Round 5 probe: parity-gated bench lines for batches of small pictures (scans below one workgroup are cut shorter), the multi-scan
# forms, and the headline; bench.py refuses timed regions in which a chunk had not converged.
for wh in "256 256 32768" "512 512 16384" "1024 768 8192" "640 480 8192" "1920 1080 4096" "3840 2160 2048"; do set -- $wh; python bench.py --no-cpu-baseline --no-extra --width $1 --height $2 --images-per-gpu $3 2>&1 | grep '^{\|Assert' | python3 -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print('$1x$2', round(d['value']), d['ms_per_step'], d['unconverged_chunk_runs'], d['parity']['ok'], d['parity']['tiled_max_abs_diff'], d['parity']['max_abs_diff'], d['parity']['t0_equal'])
except Exception: print('$1x$2', t[-300:])"; done
python tools/bench_multiscan.py 2>&1 | tail -3 | cut -c1-230
