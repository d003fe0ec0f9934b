#!/bin/bash
# One-stream pass per build with the geometry of the batch: tools/abv2.sh "lib1 lib2 ..." [bench args]
LIBS=$1; shift
show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), d['ms_per_step'], 'sub_bytes', d['config'].get('subsequence_bytes'), {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})" $1; }
for L in $LIBS; do
  MJX_STREAMS=1 MJX_BENCH_IGNORE_STATUS=1 MJX_LIB=$PWD/$L timeout 600 python3 bench.py --no-cpu-baseline --no-extra --no-parity "$@" 2>/dev/null | show $L
done
