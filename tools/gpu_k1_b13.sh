#!/bin/bash
# scatter with 8192-row buckets (384-byte runs at the fine levels; K2: 128 KiB of accumulators, one workgroup of 1024 threads per CU) and
# 4096-row buckets against the product's 2048 (ab/libsdfx_hip_b13.so / _b12.so = python tools/build_variant.py b13 gridencoder_bwd_binned.hip
# -DSDFX_BUCKET_LOG2=13), rounds alternating; the table gradients of every variant are compared with the oracle-checked first call of its run
TAG=${1:-b13}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp SDFX_DEV=1
for R in 1 2; do
for LIB in stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so ab/libsdfx_hip_b12.so ab/libsdfx_hip_b13.so; do
  echo "#### round $R $LIB" | tee -a $OUT/k1_b13.txt
  SDFX_LIB=$PWD/$LIB timeout 300 python tools/scatter_ab.py 2 10 2>&1 | grep -v amdgpu.ids | tee -a $OUT/k1_b13.txt | grep -v "round [0-9] {" | cut -c1-300
done; done
