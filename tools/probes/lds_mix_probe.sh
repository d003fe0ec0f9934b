#!/bin/bash
# Default streams: LDS padding of the entropy kernels so that a CU that holds their workgroups has room left for stage B's
# (run through gpurun).  Round 4: no gain -- and a write pass with one workgroup per CU is half again as slow (tools/probes/occupancy_probe.sh).
show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), d['ms_per_step'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})" "$1"; }
Q="--no-cpu-baseline --no-extra --no-parity --no-traffic"
for r in 1 2; do
  timeout 300 python3 bench.py $Q "$@" 2>/dev/null | show base
  MJX_WRITE_LDS_PAD=4096 timeout 300 python3 bench.py $Q "$@" 2>/dev/null | show "write+4K"
  MJX_WRITE_LDS_PAD=24576 timeout 300 python3 bench.py $Q "$@" 2>/dev/null | show "write+24K(1/CU)"
  MJX_WRITE_LDS_PAD=4096 MJX_SPEC_LDS_PAD=18432 timeout 300 python3 bench.py $Q "$@" 2>/dev/null | show "write+4K,spec+18K(2/CU)"
  MJX_WRITE_LDS_PAD=4096 MJX_MERGE_LDS_PAD=8192 timeout 300 python3 bench.py $Q "$@" 2>/dev/null | show "write+4K,merge+8K"
done
