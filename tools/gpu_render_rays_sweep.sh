#!/bin/bash
# Fused render kernels: rays per workgroup x sibling waves per ray, fixed + marginal fit on the GPU clock (tools/render_fit.py), two rounds
# interleaved.   gpurun -- 'bash tools/gpu_render_rays_sweep.sh <tag>'
TAG=${1:-render_rays}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export SDFX_LIB=$PWD/stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so
for rnd in 1 2; do
  for cfg in "2 1" "2 2" "2 4" "1 2" "1 4" "4 1" "4 2"; do
    set -- $cfg
    SDFX_RENDER_WAVES=$1 SDFX_RENDER_RAYS=$2 timeout 200 python tools/render_fit.py 2>/dev/null | tail -1 | tee -a $OUT/summary.txt
  done
done
