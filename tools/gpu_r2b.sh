#!/bin/bash
# Round-2 second GPU call: gather-policy microbenchmark, encode-forward variants (timing + bit-exactness), grid parity tests.
TAG=${1:-r2b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 120 tools/ubench/gather_policy.bin > $OUT/gather_policy.txt 2>&1
echo "gather_policy exit: $?" | tee -a $OUT/summary.txt
cat $OUT/gather_policy.txt | tee -a $OUT/summary.txt
for KIND in stencil uniform ray; do
  timeout 300 python tools/encode_bench.py $KIND f16 20 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
done
timeout 300 python tools/encode_bench.py stencil f32 10 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vs_reference_kernels.py -m gpu -q --timeout 300 -p no:cacheprovider -k "grid or field or fused" 2>&1 | tail -15 | tee -a $OUT/summary.txt
