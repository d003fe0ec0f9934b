# Round 5 probe: pictures of the size and quality of a vision data set (500x375 at quality 90), per-kernel times on one stream and the overlapped rate
for q in 75 90; do
MJX_STREAMS=1 python bench.py --no-cpu-baseline --no-extra --no-parity --width 500 --height 375 --quality $q --images-per-gpu 16384 2>&1 | grep '^{\|Assert' | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('q$q one stream', round(d['value']), 'Mpx/s', d['ms_per_step'], 'ms', round(16384/d['ms_per_step']), 'k images/s', d['config']['subsequence_bytes'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})"
python bench.py --no-cpu-baseline --no-extra --width 500 --height 375 --quality $q --images-per-gpu 16384 2>&1 | grep '^{\|Assert' | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('q$q default   ', round(d['value']), 'Mpx/s', d['ms_per_step'], 'ms', round(16384/d['ms_per_step']), 'k images/s', d['unconverged_chunk_runs'], d['timed_again_after_a_repair'], d['parity']['ok'])"
done
