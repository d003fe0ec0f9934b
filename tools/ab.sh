#!/bin/bash
# A/B two builds on the same GPU box: tools/ab.sh ab/libmjx_A.so ab/libmjx_B.so [bench args]
A=$1; B=$2; shift 2
for r in 1 2 3; do for L in $A $B; do
  MJX_LIB=$PWD/$L timeout 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d['value'], d['ms_per_step'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})" $L
done; done
