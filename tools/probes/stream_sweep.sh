#!/bin/bash
# Stream configurations of the context on one box (run through gpurun): tools/probes/stream_sweep.sh
show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), d['ms_per_step'])" "$1"; }
Q="--no-cpu-baseline --no-extra --no-parity --no-traffic"
for r in 1 2; do
  timeout 300 python3 bench.py $Q 2>/dev/null | show "default(3)"
  MJX_STREAMS=2 timeout 300 python3 bench.py $Q 2>/dev/null | show "streams=2,pixels-high"
  MJX_STREAMS=2 MJX_HIGH_PRIO=entropy timeout 300 python3 bench.py $Q 2>/dev/null | show "streams=2,entropy-high"
  MJX_STREAMS=2 MJX_HIGH_PRIO=none timeout 300 python3 bench.py $Q 2>/dev/null | show "streams=2,none"
  MJX_HIGH_PRIO=pixels timeout 300 python3 bench.py $Q 2>/dev/null | show "default(3),pixels-high"
  MJX_HIGH_PRIO=entropy timeout 300 python3 bench.py $Q 2>/dev/null | show "default(3),entropy(even chunks)-high"
  MJX_STREAMS=1 timeout 300 python3 bench.py $Q 2>/dev/null | show "streams=1"
done
