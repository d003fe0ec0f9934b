#!/bin/bash
# A/B of several builds on one GPU box, also measurement builds that decode garbage: tools/abx.sh "lib1 lib2 ..." [bench args]
LIBS=$1; shift
for r in 1 2; do for L in $LIBS; do
  MJX_BENCH_IGNORE_STATUS=1 MJX_LIB=$PWD/$L timeout 600 python bench.py --no-cpu-baseline --no-extra --no-parity "$@" 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), d['ms_per_step'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()}, d.get('parity',{}).get('ok'))" $L
done; done
