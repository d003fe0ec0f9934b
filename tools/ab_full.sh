#!/bin/bash
# full headline step (entropy + stage B overlapped, default streams) for several builds on one box: tools/ab_full.sh "lib1 lib2" [extra bench args]
for r in 1 2; do for L in $1; do
  MJX_LIB=$PWD/$L timeout 900 python3 bench.py --no-cpu-baseline --no-extra --no-parity --steps 10 $2 2>/dev/null | grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), d['ms_per_step'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})" $L
done; done
