#!/bin/bash
# One GPU-box call that produces everything profiles/<round>_* is built from (run through gpurun from the repo root):
#   tools/profile_round.sh r03
# the default bench line (the library's default streams: three, parity gate, extra configs, CPU baseline), rocprofv3 kernel stats of the 4K run with one
# and with the default streams, HBM traffic counters (FETCH_SIZE / WRITE_SIZE in separate passes, one launch per kernel) and the SQ
# counters in two passes (one stream: per-kernel counters are only meaningful without overlap).  Every command under a timeout of its own.
R=$1
O=gpurun_out/$R; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --no-extra --no-parity"
timeout 900 python3 bench.py > $O/bench_2048x4K.json 2> $O/bench_2048x4K.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats2 -o out --output-format csv -- python3 bench.py $Q --steps 3 --warmup 1 > $O/stats2.log 2>&1
export MJX_STREAMS=1
timeout 300 python3 bench.py $Q > $O/bench_2048x4K_1stream.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats1 -o out --output-format csv -- python3 bench.py $Q --steps 3 --warmup 1 > $O/stats1.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c -d $O/pmc_$c -o out --output-format csv -- python3 bench.py $Q --steps 1 --warmup 0 --images-per-gpu 256 > $O/pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_sqA -o out --output-format csv -- python3 bench.py $Q --steps 1 --warmup 0 --images-per-gpu 256 > $O/pmc_sqA.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR -d $O/pmc_sqB -o out --output-format csv -- python3 bench.py $Q --steps 1 --warmup 0 --images-per-gpu 256 > $O/pmc_sqB.log 2>&1
unset MJX_STREAMS
timeout 400 python3 tools/single_image_times.py > $O/single_images.txt 2>&1
timeout 400 python3 tools/e2e_from_files.py 512 0 > $O/e2e_512.txt 2>&1
timeout 400 python3 tools/e2e_from_files.py 2048 0 > $O/e2e_2048.txt 2>&1
timeout 400 python3 tools/e2e_from_files.py 512 0 0 > $O/e2e_512_host_destuff.txt 2>&1
timeout 400 python3 tools/e2e_from_files.py 2048 0 0 > $O/e2e_2048_host_destuff.txt 2>&1
timeout 400 python3 tools/e2e_from_files.py 512 0 1 > $O/e2e_512_device_destuff.txt 2>&1
timeout 400 python3 tools/e2e_from_files.py 2048 0 1 > $O/e2e_2048_device_destuff.txt 2>&1
timeout 400 python3 tools/batch_size_sweep.py > $O/batch_size_sweep.txt 2>&1
timeout 900 bash tools/probes/small_parity.sh > $O/small_pictures_and_multiscan.txt 2>&1       # parity-gated lines: small pictures, 1080p, 4K, the multi-scan forms
timeout 400 bash tools/probes/subsampling_sweep.sh > $O/subsampling_1080p_1stream.txt 2>&1    # 4:2:0 / 4:2:2 / 4:4:4 / grey, per-kernel times
timeout 400 python3 tools/collect_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/traffic.json 256 3840 2160 > $O/traffic.log 2>&1
timeout 400 python3 tools/pmc_summary.py $O/pmc_sqA/*counter_collection.csv $O/pmc_sqB/*counter_collection.csv > $O/pmc_sq_summary.txt 2>&1
cp $O/stats2/*kernel_stats.csv $O/kernel_stats_2048x4K_default_streams.csv 2>/dev/null
cp $O/stats1/*kernel_stats.csv $O/kernel_stats_2048x4K_1stream.csv 2>/dev/null
rm -rf $O/pmc_*/ $O/stats1 $O/stats2
ls -la $O | head -30; tail -1 $O/bench_2048x4K.json | cut -c1-600
