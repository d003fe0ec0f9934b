# Round 5: the chunk split of a 2048 x 4K step with the single-decode kernels (3 : 1 built in, equal halves, three, four chunks, scan-byte targets), two repeats.
show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', round(d['value']), d['ms_per_step'], d['config']['chunks_per_step'])" "$1"; }
for r in 1 2; do
for c in 0 1024 683 512 1280; do
  timeout 600 python3 bench.py --no-cpu-baseline --no-extra --no-parity --chunk-images $c 2>/dev/null | show "chunk-images=$c"
done
for m in 1024 2048; do MJX_CHUNK_SCAN_MB=$m timeout 600 python3 bench.py --no-cpu-baseline --no-extra --no-parity 2>/dev/null | show "scan_mb=$m"; done
done
