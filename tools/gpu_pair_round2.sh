#!/bin/bash
# pair plan: cost candidates + tiles-per-workgroup sweep on the devtools library, then the register / overlap variants (ab/*.so), same box
TAG=${1:-pair2}; VIEWS=${2:-2}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp SDFX_DEV=1
C1="0.001,0.001,0.001,0.001,0.001,0.001,0.001,0.001,25.6,24.1,25.2,24.8,27.5,31.6,34.7,36.8"
C2="0.001,0.001,0.001,0.001,0.001,0.001,0.001,0.001,26.5,24.5,25.5,25.0,27.5,31.0,34.0,36.0"
SDFX_LIB=$PWD/stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so PAIR_TPW="2,4,8" PAIR_COSTS="$C1;$C2" timeout 900 python tools/pair_ab.py $VIEWS 10 2>&1 | tee $OUT/pair_ab_costs.txt | grep -v -E '^   XCD [0-9]|round' | cut -c1-260
for V in pairw7 pairov6 pairov5; do
  echo "#### variant $V" | tee -a $OUT/variants.txt
  SDFX_LIB=$PWD/ab/libsdfx_hip_$V.so PAIR_TPW="4" timeout 600 python tools/pair_ab.py $VIEWS 10 2>&1 | tee -a $OUT/variants.txt | grep -E 'min |spread|wg,' | cut -c1-260
done
