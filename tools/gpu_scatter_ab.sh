#!/bin/bash
# scatter K1: interleaved fine / coarse tiles per XCD (SDFX_GRIDBWD_INTERLEAVE) on the devtools library and on the tiles-per-workgroup variants
TAG=${1:-sc1}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp SDFX_DEV=1
for LIB in stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so ab/libsdfx_hip_bintpw2.so ab/libsdfx_hip_bintpw4.so; do
  echo "#### $LIB" | tee -a $OUT/scatter_ab.txt
  SDFX_LIB=$PWD/$LIB timeout 600 python tools/scatter_ab.py 2 10 SDFX_GRIDBWD_INTERLEAVE=0,1 2>&1 | grep -v amdgpu.ids | tee -a $OUT/scatter_ab.txt | grep -v round | cut -c1-300
done
