#!/bin/bash
# scatter K1: does its time follow its vector instruction count? n dummy full-rate VALU instructions per thread and level
# (ab/libsdfx_hip_k1pad<n>.so = python tools/build_variant.py k1pad<n> gridencoder_bwd_binned.hip -DSDFX_K1_PAD=<n>), rounds alternating
TAG=${1:-k1pad}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp SDFX_DEV=1
for R in 1 2; do
for LIB in stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so ab/libsdfx_hip_k1pad64.so ab/libsdfx_hip_k1pad128.so; do
  echo "#### round $R $LIB" | tee -a $OUT/k1_pad.txt
  SDFX_LIB=$PWD/$LIB timeout 300 python tools/scatter_ab.py 2 10 2>&1 | grep -v amdgpu.ids | tee -a $OUT/k1_pad.txt | grep -v round | cut -c1-300
done; done
