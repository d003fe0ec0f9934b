show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), d['ms_per_step'], d['config'].get('chunks_per_step'))" "$1"; }
Q="--no-cpu-baseline --no-extra --no-parity --no-traffic"
for r in 1 2; do for c in 1100 1200 1300 1400 1500 1600; do
  MJX_LIB=$PWD/ab/libmjx_long2.so timeout 300 python3 bench.py $Q --chunk-images $c 2>/dev/null | show "long2 chunk$c default"
done; done
