#!/bin/bash
# Stage B alone (the config-4 sweep: 4096 x 1080p, --stages pixels) for several builds on one GPU box: tools/ab_stageb.sh "ab/libmjx_a.so ab/libmjx_b.so"
# (measurement builds that decode garbage are fine: statuses and parity are not looked at)
for r in 1 2; do for L in $1; do
  MJX_BENCH_IGNORE_STATUS=1 MJX_LIB=$PWD/$L timeout 600 python3 bench.py --no-cpu-baseline --no-extra --no-parity --stages pixels --width 1920 --height 1080 --images-per-gpu 4096 --steps 5 2>/dev/null | grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], 'pixels-only 4096x1080p', round(d['value']), d['ms_per_step'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})" $L
done; done
