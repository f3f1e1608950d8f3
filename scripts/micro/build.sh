#!/bin/bash
# usage: build.sh <out> [extra hipcc flags]
out=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=1000000 -I /root/repo/industrial_nnmpc_2021_amd/csrc -DASM_WG_MICRO \
  "$@" /root/repo/scripts/micro/lambda_micro.hip -o $out
