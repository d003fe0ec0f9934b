# Round 5 probe: entropy workgroups of 128 / 256 lanes (-DMJX_HUFF_WG) against 512 on batches of small pictures
for L in jpeg-rust_amd/libmjx.so ab/libmjx_wg256.so ab/libmjx_wg128.so; do
for wh in "256 256 32768" "512 512 16384" "500 375 16384"; do set -- $wh; echo -n "$L $1x$2 "; MJX_FIX_PASSES=12 MJX_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-extra --no-parity --width $1 --height $2 --images-per-gpu $3 2>&1 | grep '^{\|Assert' | python3 -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(round(d['value']), d['ms_per_step'], d['config']['subsequence_bytes'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})
except Exception: print(t[-200:])"; done; done
