show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), d['ms_per_step'], d['config'].get('chunks_per_step'))" "$1"; }
Q="--no-cpu-baseline --no-extra --no-parity --no-traffic"
for L in g512 long2; do for c in 2048 1366 1024 683; do
  MJX_LIB=$PWD/ab/libmjx_$L.so timeout 300 python3 bench.py $Q --chunk-images $c 2>/dev/null | show "$L chunk$c default"
  MJX_STREAMS=1 MJX_LIB=$PWD/ab/libmjx_$L.so timeout 300 python3 bench.py $Q --chunk-images $c 2>/dev/null | show "$L chunk$c one-stream"
done; done
