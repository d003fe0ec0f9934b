# Round 5 probe: the generic stage-B form (4:4:4, 4:2:2, grey) beside the 4:2:0 one, one stream, per-kernel times; MJX_LIB: a variant build
for sub in 420 422 444 gray; do for wh in "1920 1080 2048"; do set -- $wh; MJX_STREAMS=1 python bench.py --no-cpu-baseline --no-extra --subsampling $sub --width $1 --height $2 --images-per-gpu $3 2>&1 | grep '^{\|Error' | python3 -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print('$sub $1x$2', round(d['value']), d['ms_per_step'], d['unconverged_chunk_runs'], d['parity']['ok'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})
except Exception: print('$sub', t[-300:])"; done; done
