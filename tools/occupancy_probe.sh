show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), d['ms_per_step'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})" "$1"; }
Q="--no-cpu-baseline --no-extra --no-parity --no-traffic"
export MJX_STREAMS=1
timeout 300 python3 bench.py $Q 2>/dev/null | show base
MJX_WRITE_LDS_PAD=4096 timeout 300 python3 bench.py $Q 2>/dev/null | show "write 1wg/cu"
MJX_SPEC_LDS_PAD=6144 timeout 300 python3 bench.py $Q 2>/dev/null | show "spec 3wg/cu(41K)"
MJX_SPEC_LDS_PAD=20480 timeout 300 python3 bench.py $Q 2>/dev/null | show "spec 2wg/cu(55K)"
