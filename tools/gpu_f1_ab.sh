#!/bin/bash
# SURVEY section 8 f1, forward half, on THIS round's kernel: the hinted encode forward level-major (the product's plan) against
# SDFX_GRID_PLAN=sample_major — every XCD takes 1/8 of the tiles and evaluates ALL 16 levels of a tile before the next tile, the table
# access pattern of an encode fused into the field MLP. Three interleaved rounds; outputs are bit-identical.
TAG=${1:-f1_ab}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export SDFX_LIB=$PWD/stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so
for rnd in 1 2 3; do
  for plan in level_major sample_major; do
    for kind in stencil ray; do
      echo -n "round $rnd $plan: " | tee -a $OUT/summary.txt
      SDFX_GRID_PLAN=$plan timeout 200 python tools/encode_bench.py $kind f16 5 1,1,1 2>/dev/null | tail -1 | tee -a $OUT/summary.txt
    done
  done
done
