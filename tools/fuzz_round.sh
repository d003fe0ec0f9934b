#!/bin/bash
# The randomised checks of a round on one GPU box (run through gpurun from the repo root): tools/fuzz_round.sh <out.log> [images per sweep]
# big-noisy and huge-image checks, fuzz sweeps (default; every picture through the single-decode kernels -- MJX_EMIT_MIN_SUB_BITS=256
# makes the emitting first decode take short subsequences too, which the random sizes of the sweep otherwise never reach --; the same
# with no head room, so that prefixes that grow hand their pictures to the two-pass kernels; two-pass only; linear stream + poisoned
# scratch; REF_COMPAT; device de-stuffing is drawn at random inside every sweep), hostile inputs, the big multi-scan picture.
OUT=$1; N=${2:-600}
{
echo "== $(python3 -c 'import __graft_entry__ as g; print(g.load_package().lib().mjx_version().decode() if hasattr(g.load_package().lib().mjx_version, "__call__") else "")' 2>/dev/null) =="
echo "-- big_noisy / huge_image"; timeout 900 python3 tools/big_noisy_check.py 2>&1 | tail -6; timeout 900 python3 tools/huge_image_check.py 2>&1 | tail -12
echo "-- fuzz: default";                                   timeout 1200 python3 tools/fuzz_parity.py 81 $N 2>&1 | tail -2
echo "-- fuzz: MJX_EMIT_MIN_SUB_BITS=256 (single decode for every picture of one scan)"; MJX_EMIT_MIN_SUB_BITS=256 timeout 1200 python3 tools/fuzz_parity.py 82 $N 2>&1 | tail -2
echo "-- fuzz: MJX_EMIT_MIN_SUB_BITS=256 MJX_EMIT_WARM_BITS=0 MJX_EMIT_CP_BITS=256"; MJX_EMIT_MIN_SUB_BITS=256 MJX_EMIT_WARM_BITS=0 MJX_EMIT_CP_BITS=256 timeout 1200 python3 tools/fuzz_parity.py 83 $N 2>&1 | tail -2
echo "-- fuzz: MJX_EMIT_MIN_SUB_BITS=256 MJX_EMIT_HEAD=0 (no head room: fall-back path)"; MJX_EMIT_MIN_SUB_BITS=256 MJX_EMIT_WARM_BITS=0 MJX_EMIT_HEAD=0 timeout 1200 python3 tools/fuzz_parity.py 84 $N 2>&1 | tail -2
echo "-- fuzz: MJX_EMIT_MIN_SUB_BITS=256 MJX_POISON=165 MJX_STREAMS=1"; MJX_EMIT_MIN_SUB_BITS=256 MJX_POISON=165 MJX_STREAMS=1 timeout 1200 python3 tools/fuzz_parity.py 85 $N 2>&1 | tail -2
echo "-- fuzz: FUZZ_THROUGHPUT_PLAN=1 (the cut of a large batch: scans below one workgroup cut to fill it)"; FUZZ_THROUGHPUT_PLAN=1 timeout 1200 python3 tools/fuzz_parity.py 89 $N 2>&1 | tail -2
echo "-- fuzz: MJX_SINGLE_DECODE=0";                       MJX_SINGLE_DECODE=0 timeout 1200 python3 tools/fuzz_parity.py 86 $N 2>&1 | tail -2
echo "-- fuzz: MJX_STREAM_LINEAR=1 MJX_POISON=90";         MJX_STREAM_LINEAR=1 MJX_POISON=90 timeout 1200 python3 tools/fuzz_parity.py 87 $N 2>&1 | tail -2
echo "-- fuzz: REF_COMPAT, MJX_EMIT_MIN_SUB_BITS=256";     MJX_EMIT_MIN_SUB_BITS=256 timeout 1200 python3 tools/fuzz_parity.py 88 $N - ref 2>&1 | tail -2
echo "-- hostile";                                          timeout 900 python3 tools/stress_hostile.py 2>&1 | tail -2
echo "-- hostile, MJX_EMIT_MIN_SUB_BITS=256";              MJX_EMIT_MIN_SUB_BITS=256 timeout 900 python3 tools/stress_hostile.py 2>&1 | tail -2
echo "-- fuzz_planar (multi-scan twins read from their scans' streams)"; timeout 1200 python3 tools/fuzz_planar.py 7 120 2>&1 | tail -2
echo "-- big_multiscan";                                    timeout 900 python3 tools/big_multiscan_check.py 2>&1 | tail -3
} > $OUT 2>&1
