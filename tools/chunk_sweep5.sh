show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), d['ms_per_step'], d['config'].get('chunks_per_step'))" "$1"; }
Q="--no-cpu-baseline --no-extra --no-parity --no-traffic"
for c in 0 2000 1800 1536 1366 1024; do
  timeout 300 python3 bench.py $Q --quality 50 --chunk-images $c 2>/dev/null | show "q50 chunk$c"
done
for c in 0 3600 3072 2731 2048; do
  timeout 300 python3 bench.py $Q --width 1920 --height 1080 --images-per-gpu 4096 --chunk-images $c 2>/dev/null | show "1080p chunk$c"
done
