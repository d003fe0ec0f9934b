#!/bin/bash
# stage-B-only sweep (--stages pixels) of several builds / settings on one box: tools/abp.sh "lib[:ENV=val,ENV=val] ..." [bench args]
SPECS=$1; shift
for r in 1 2; do for S in $SPECS; do
  L=${S%%:*}; E=""; [ "$S" != "$L" ] && E=${S#*:}
  env $(echo $E | tr ',' ' ') MJX_LIB=$PWD/$L python3 bench.py --no-cpu-baseline --no-extra --no-parity --stages pixels "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), d['ms_per_step'], d['parity']['tiled_max_abs_diff'])" $S
done; done
