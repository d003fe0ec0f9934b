#!/bin/bash
# Kernel chunks per step (run through gpurun): by picture count (--chunk-images) and by scan bytes (MJX_CHUNK_SCAN_MB), default streams.
#   tools/probes/chunk_sweep.sh ["bench args" ...]      e.g.  tools/probes/chunk_sweep.sh "" "--quality 90" "--width 1920 --height 1080 --images-per-gpu 4096"
show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), d['ms_per_step'], 'chunks', d['config'].get('chunks_per_step'))" "$1"; }
Q="--no-cpu-baseline --no-extra --no-parity --no-traffic"
[ $# -eq 0 ] && set -- ""
for a in "$@"; do
  for m in 1024 1280 1536 2048; do MJX_CHUNK_SCAN_MB=$m timeout 300 python3 bench.py $Q $a 2>/dev/null | show "[$a] scan_mb=$m"; done
  for c in 2048 1536 1024 683; do timeout 300 python3 bench.py $Q $a --chunk-images $c 2>/dev/null | show "[$a] chunk_images=$c"; done
done
