#!/bin/bash
# As tools/abe.sh (several environments on one box, default streams only), for another workload: tools/abe_cfg.sh "ENV1" "ENV2" -- --width 1920 --height 1080 --images-per-gpu 4096
ENVS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done; [ "$1" = "--" ] && shift
show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', round(d['value']), d['ms_per_step'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()}, d.get('parity',{}).get('tiled_max_abs_diff'))" "$1"; }
for r in 1 2; do for E in "${ENVS[@]}"; do
  env $E timeout 600 python3 bench.py --no-cpu-baseline --no-extra --no-parity "$@" 2>/dev/null | show "$E"
done; done
