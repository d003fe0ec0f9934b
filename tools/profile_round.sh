#!/bin/bash
# One GPU-box call that produces everything profiles/<round>_* is built from (run through gpurun from the repo root):
#   tools/profile_round.sh r01f
# bench lines (4K, 4K two streams, 1080p, 1080p stage B), rocprofv3 kernel stats of the 4K run, HBM traffic counters
# (FETCH_SIZE / WRITE_SIZE in separate passes, one launch per kernel) and the SQ counters in three passes.
R=$1
O=gpurun_out/$R; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 bench.py > $O/bench_2048x4K.json 2> $O/bench_2048x4K.err
python3 bench.py --no-cpu-baseline --streams 2 > $O/bench_2048x4K_2streams.json 2>/dev/null
python3 bench.py --no-cpu-baseline --width 1920 --height 1080 --images-per-gpu 4096 > $O/bench_4096x1080p.json 2>/dev/null
python3 bench.py --no-cpu-baseline --width 1920 --height 1080 --images-per-gpu 4096 --stages pixels > $O/bench_4096x1080p_stageB.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/stats -o out --output-format csv -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/pmc_$c -o out --output-format csv -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --images-per-gpu 256 > $O/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_sqA -o out --output-format csv -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --images-per-gpu 256 > $O/pmc_sqA.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR -d $O/pmc_sqB -o out --output-format csv -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --images-per-gpu 256 > $O/pmc_sqB.log 2>&1
python3 tools/collect_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/traffic.json 256 > /dev/null
python3 tools/pmc_summary.py $O/pmc_sqA/*counter_collection.csv $O/pmc_sqB/*counter_collection.csv > $O/pmc_sq_summary.txt 2>&1
cp $O/stats/*kernel_stats.csv $O/kernel_stats_2048x4K.csv 2>/dev/null
ls -la $O | head -30; tail -1 $O/bench_2048x4K.json | cut -c1-400
