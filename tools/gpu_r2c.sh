#!/bin/bash
TAG=${1:-r2c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 120 tools/ubench/gather_policy.bin > $OUT/gather_policy.txt 2>&1
echo "gather_policy exit: $?" | tee -a $OUT/summary.txt
cat $OUT/gather_policy.txt | tee -a $OUT/summary.txt
for KIND in stencil7 stencil; do
  timeout 300 python tools/encode_bench.py $KIND f16 20 0,0 1,0 1,1 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
done
STENCIL_ORDER=stencil PER_LEVEL=1 timeout 300 python tools/gridbwd_bench.py 10 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
STENCIL_ORDER=sample PER_LEVEL=1 timeout 300 python tools/gridbwd_bench.py 10 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
