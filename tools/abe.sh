#!/bin/bash
# Several environments on one GPU box, one stream and the default streams, the library as built:
#   tools/abe.sh "MJX_SINGLE_DECODE=0" "MJX_SINGLE_DECODE=1 MJX_EMIT_WARM_BITS=2048" -- [bench args]
ENVS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done; [ "$1" = "--" ] && shift
show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', sys.argv[2], round(d['value']), d['ms_per_step'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()}, d.get('parity',{}).get('tiled_max_abs_diff'))" "$1" "$2"; }
for r in 1 2; do for E in "${ENVS[@]}"; do
  env $E MJX_STREAMS=1 timeout 600 python3 bench.py --no-cpu-baseline --no-extra --no-parity "$@" 2>/dev/null | show "$E" one-stream
  env $E timeout 600 python3 bench.py --no-cpu-baseline --no-extra --no-parity "$@" 2>/dev/null | show "$E" default
done; done
