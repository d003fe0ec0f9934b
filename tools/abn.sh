#!/bin/bash
# Compare N builds on the same GPU box: tools/abn.sh "lib1 lib2 ..." [bench args]; env vars of the caller pass through
LIBS=$1; shift
for r in 1 2; do for L in $LIBS; do
  MJX_LIB=$PWD/$L timeout 600 python bench.py --no-cpu-baseline --no-extra --no-parity "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), d['ms_per_step'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})" $L
done; done
