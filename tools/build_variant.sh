#!/bin/bash
# Build a variant of libmjx.so with extra -D flags for A/B runs: tools/build_variant.sh NAME -DMJX_AHEAD=2 ...
# (objects of the regular build are reused for everything but the kernels and the API)
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/jpeg-rust_amd/csrc; B=$R/jpeg-rust_amd/build; mkdir -p $R/ab /tmp/mjxv_$NAME
for s in mjx_kernels.hip mjx_api.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c $C/$s -o /tmp/mjxv_$NAME/$s.o -I$R/include -I$C
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/libmjx_$NAME.so /tmp/mjxv_$NAME/mjx_kernels.hip.o /tmp/mjxv_$NAME/mjx_api.hip.o $B/mjx_lut.cpp.o $B/mjx_parse.cpp.o $B/mjx_plan.cpp.o
echo ab/libmjx_$NAME.so
