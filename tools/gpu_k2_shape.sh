#!/bin/bash
# scatter K2 (k_grid_bwd_reduce_fixed): threads per workgroup (-DSDFX_REDUCE_THREADS=256 / 1024; product 512) and loads in flight per thread
# (-DSDFX_REDUCE_UNROLL=4 / 16; product 8), each built with tools/build_variant.py <name> gridencoder_bwd_binned.hip <flag>; rounds alternating
TAG=${1:-k2shape}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp SDFX_DEV=1
for R in 1 2; do
for LIB in stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so ab/libsdfx_hip_rt256.so ab/libsdfx_hip_rt1024.so ab/libsdfx_hip_un4.so ab/libsdfx_hip_un16.so; do
  echo "#### round $R $LIB" | tee -a $OUT/k2_shape.txt
  SDFX_LIB=$PWD/$LIB timeout 300 python tools/scatter_ab.py 2 10 2>&1 | grep -v amdgpu.ids | tee -a $OUT/k2_shape.txt | grep -v "round [0-9] {" | cut -c1-300
done; done
