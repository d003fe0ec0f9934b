show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), d['ms_per_step'], d['config'].get('chunks_per_step'))" "$1"; }
Q="--no-cpu-baseline --no-extra --no-parity --no-traffic"
for a in "" "--quality 90" "--quality 50"; do for m in 1024 1280 1536 2048 3072; do
  MJX_CHUNK_SCAN_MB=$m timeout 300 python3 bench.py $Q $a 2>/dev/null | show "[$a] scan_mb=$m default"
done; done
