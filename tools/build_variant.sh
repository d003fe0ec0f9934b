#!/bin/bash
# Build a variant of libmjx.so with extra -D flags for A/B runs: tools/build_variant.sh NAME -DMJX_HUFF_WG=1024 ...
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/jpeg-rust_amd/csrc; mkdir -p $R/ab /tmp/mjxv_$NAME
rm -f /tmp/mjxv_$NAME/*.o
pids=""
for s in mjx_kernels.hip mjx_api.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c $C/$s -o /tmp/mjxv_$NAME/$s.o -I$R/include -I$C &
  pids="$pids $!"
done
for s in mjx_lut.cpp mjx_parse.cpp mjx_plan.cpp mjx_pool.cpp; do
  g++ -O2 -std=c++17 -fPIC "$@" -c $C/$s -o /tmp/mjxv_$NAME/$s.o -I$R/include -I$C
done
for p in $pids; do wait $p; done       # (a failed compile stops the script: set -e)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/libmjx_$NAME.so /tmp/mjxv_$NAME/*.o
echo ab/libmjx_$NAME.so
