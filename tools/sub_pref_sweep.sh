#!/bin/bash
# The length choose_subseq_bits aims at (MJX_SUB_PREF, bits), one stream and default streams: tools/sub_pref_sweep.sh "prefs" [bench args]
PREFS=$1; shift
show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], round(d['value']), d['ms_per_step'], 'sub_bytes', d['config'].get('subsequence_bytes'), {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})" $1 $2; }
Q="--no-cpu-baseline --no-extra --no-parity --no-traffic"
for p in $PREFS; do
  MJX_SUB_PREF=$p MJX_STREAMS=1 timeout 300 python3 bench.py $Q "$@" 2>/dev/null | show pref$p one-stream
  MJX_SUB_PREF=$p timeout 300 python3 bench.py $Q "$@" 2>/dev/null | show pref$p default
done
