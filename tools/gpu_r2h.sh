#!/bin/bash
TAG=${1:-r2h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -60 | cut -c1-260 > $OUT/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" | tee -a $OUT/summary.txt
tail -60 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
timeout 600 python tools/infer_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
