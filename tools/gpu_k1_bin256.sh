#!/bin/bash
# scatter K1 with 256-sample tiles (ab/libsdfx_hip_bin256.so = python tools/build_variant.py bin256 gridencoder_bwd_binned.hip
# -DSDFX_BIN_THREADS=256 -DSDFX_MAX_BUCKETS=256) against the 512-sample tiles of the devtools library, rounds alternating
TAG=${1:-bin256}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp SDFX_DEV=1
for R in 1 2; do
for LIB in stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so ab/libsdfx_hip_bin256.so; do
  echo "#### round $R $LIB" | tee -a $OUT/k1_bin256.txt
  SDFX_LIB=$PWD/$LIB timeout 300 python tools/scatter_ab.py 2 10 2>&1 | grep -v amdgpu.ids | tee -a $OUT/k1_bin256.txt | grep -v round | cut -c1-300
done; done
