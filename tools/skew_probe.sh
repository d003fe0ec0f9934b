show() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), d['ms_per_step'], {k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})" "$1"; }
Q="--no-cpu-baseline --no-extra --no-parity --no-traffic --chunk-images 1024"
export MJX_STREAMS=1 MJX_BENCH_IGNORE_STATUS=1
for sz in 512 640 704; do for sk in 0 16 61; do
  MJX_SKEW=$sk MJX_LIB=$PWD/ab/libmjx_k$sz.so timeout 300 python3 bench.py $Q 2>/dev/null | show "s$sz skew$sk"
done; done
