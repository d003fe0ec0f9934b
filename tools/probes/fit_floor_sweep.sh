# Round 5 probe: the shortest cut of scans below one workgroup (MJX_FIT_SHORT=<bits>), default streams
for fl in 512 1024 1536 2048 3072; do for wh in "256 256 32768" "512 512 16384" "640 480 8192" "1024 768 8192"; do set -- $wh; echo -n "floor=$fl $1x$2 "; MJX_FIX_PASSES=12 MJX_FIT_SHORT=$fl python bench.py --no-cpu-baseline --no-extra --no-parity --width $1 --height $2 --images-per-gpu $3 2>&1 | grep '^{\|Assert' | python3 -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(round(d['value']), d['ms_per_step'], d['config']['subsequence_bytes'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})
except Exception: print(t[15:70])"; done; done
