# Round 5: further seeds of the randomised sweeps on the final build (beyond tools/fuzz_round.sh's)
for s in 101 102 103; do timeout 900 python3 tools/fuzz_parity.py $s 900 2>&1 | tail -1; done
for s in 104 105; do FUZZ_THROUGHPUT_PLAN=1 timeout 900 python3 tools/fuzz_parity.py $s 900 2>&1 | tail -1; done
for s in 106 107; do MJX_EMIT_MIN_SUB_BITS=256 timeout 900 python3 tools/fuzz_parity.py $s 900 2>&1 | tail -1; done
MJX_PLANAR_DIRECT=0 timeout 900 python3 tools/fuzz_parity.py 108 600 2>&1 | tail -1
timeout 900 python3 tools/fuzz_parity.py 109 600 - ref 2>&1 | tail -1
for s in 11 12 13 14; do timeout 900 python3 tools/fuzz_planar.py $s 150 2>&1 | tail -1; done
