#!/bin/bash
# What the entropy kernels lose when LDS padding leaves them fewer workgroups per CU (run through gpurun): one stream, per-kernel ms.
# (Round 4: the write pass takes 11.96 instead of 7.81 ms with one workgroup per CU instead of two -- a pad of 4 KB still leaves two --;
# the counting pass 3.25 / 3.25 / 3.55 ms at four / three / two.)
show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), d['ms_per_step'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})" "$1"; }
Q="--no-cpu-baseline --no-extra --no-parity --no-traffic"
export MJX_STREAMS=1
timeout 300 python3 bench.py $Q 2>/dev/null | show base
MJX_WRITE_LDS_PAD=4096 timeout 300 python3 bench.py $Q 2>/dev/null | show "write +4K (still 2 wg/cu)"
MJX_WRITE_LDS_PAD=16384 timeout 300 python3 bench.py $Q 2>/dev/null | show "write +16K (1 wg/cu)"
MJX_SPEC_LDS_PAD=6144 timeout 300 python3 bench.py $Q 2>/dev/null | show "spec 3wg/cu(41K)"
MJX_SPEC_LDS_PAD=20480 timeout 300 python3 bench.py $Q 2>/dev/null | show "spec 2wg/cu(55K)"
