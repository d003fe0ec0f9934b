# Round 5 probe: 4K at quality 50 cut into long subsequences (ab/libmjx_long1.so = -DMJX_LONG_SCAN_HALF_WGS=1) against the built-in short cut, single decode and two passes.
show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', round(d['value']), d['ms_per_step'], d['config']['subsequence_bytes'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})" "$1"; }
for r in 1 2; do for L in cur long1; do
  MJX_LIB=$PWD/ab/libmjx_$L.so timeout 600 python3 bench.py --no-cpu-baseline --no-extra --no-parity --quality 50 2>/dev/null | show "q50 $L"
  MJX_SINGLE_DECODE=0 MJX_LIB=$PWD/ab/libmjx_$L.so timeout 600 python3 bench.py --no-cpu-baseline --no-extra --no-parity --quality 50 2>/dev/null | show "q50 $L two-pass"
done; done
