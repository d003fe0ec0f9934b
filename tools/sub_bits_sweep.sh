#!/bin/bash
# Write / counting pass against the subsequence length, every picture at the same length (MJX_SUB_BITS; a build with
# -DMJX_SUBSEQ_BYTES=1024 so that lengths up to 16384 bits are allowed): tools/sub_bits_sweep.sh LIB "bits ..." [bench args]
LIB=$1; BITS=$2; shift; shift
show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), d['ms_per_step'], 'sub_bytes', d['config'].get('subsequence_bytes'), {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})" $1; }
export MJX_STREAMS=1 MJX_BENCH_IGNORE_STATUS=1 MJX_LIB=$PWD/$LIB
for b in $BITS; do
  MJX_SUB_BITS=$b timeout 300 python3 bench.py --no-cpu-baseline --no-extra --no-parity --no-traffic "$@" 2>/dev/null | show bits$b
done
