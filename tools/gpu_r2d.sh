#!/bin/bash
TAG=${1:-r2d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
SDFX_GRID_PLAN_DEBUG=1 timeout 300 python tools/encode_bench.py stencil f16 1 1,-1,-1,-1,1 2>&1 | grep "grid plan" | sort -u | head -10 | tee -a $OUT/summary.txt
# impl,interleave,points,balance,hint
timeout 600 python tools/encode_bench.py stencil f16 20 0,-1,-1,-1,0 1,0,1,0,0 1,1,1,0,0 1,1,2,0,0 1,1,4,0,0 1,1,1,0,1 1,1,2,0,1 1,1,4,0,1 1,1,1,1,1 1,1,2,1,1 1,1,4,1,1 1,0,2,1,1 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
timeout 300 python tools/encode_bench.py uniform f16 10 0,-1,-1,-1,0 1,0,1,0,0 1,0,2,0,0 1,0,4,0,0 1,1,2,0,0 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
timeout 300 python tools/encode_bench.py ray f16 20 0,-1,-1,-1,0 1,0,1,0,0 1,1,2,0,0 1,1,2,1,1 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
timeout 300 python tools/encode_bench.py stencil f32 10 0,-1,-1,-1,0 1,0,1,0,0 1,0,2,0,0 1,0,2,1,1 1,1,2,1,1 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vs_reference_kernels.py -m gpu -q --timeout 300 -p no:cacheprovider -k "grid or field or fused" 2>&1 | tail -15 | tee -a $OUT/summary.txt
