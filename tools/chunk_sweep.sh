show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), d['ms_per_step'])" $1; }
for r in 1 2; do for c in 0 512 342 256 128; do
  timeout 300 python3 bench.py --no-cpu-baseline --no-extra --no-parity --no-traffic --chunk-images $c 2>/dev/null | show chunk$c
done; done
