#!/bin/bash
# What the emitting pass spends on its parts: measurement builds (garbage out, MJX_EXP_NO_FALLBACK keeps the pictures on the path), one stream.
#   tools/ab_emit.sh "ab/libmjx_cur.so ab/libmjx_enoblk.so ..." ["ENV ..."]
for r in 1 2; do for L in $1; do
  env ${2:-A=1} MJX_EXP_NO_FALLBACK=1 MJX_STREAMS=1 MJX_BENCH_IGNORE_STATUS=1 MJX_LIB=$PWD/$L timeout 600 python3 bench.py --no-cpu-baseline --no-extra --no-parity 2>/dev/null | grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], round(d['value']), d['ms_per_step'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})" $L "${2:-}"
done; done
