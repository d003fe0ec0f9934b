#!/bin/bash
# one workload under several environment settings on one box: tools/ab_cfg.sh "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...   (round 6; "-" = no setting)
ARGS=$1; shift
for r in 1 2; do for E in "$@"; do
  [ "$E" = "-" ] && EE="" || EE="$E"
  env $EE timeout 900 python3 bench.py --no-cpu-baseline --no-extra --no-parity --steps 10 $ARGS 2>/dev/null | grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', round(d['value']), d['ms_per_step'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})" "$E"
done; done
