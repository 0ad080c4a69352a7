#!/bin/bash
TAG=${1:-r2n}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_render.py tests/test_gpu_trainer.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -5 | cut -c1-200 | tee -a $OUT/summary.txt
timeout 300 python tools/render_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
for W in 1 2; do
  echo "SDFX_FIELD_BWD_WAVES=$W" | tee -a $OUT/summary.txt
  SDFX_FIELD_BWD_WAVES=$W timeout 300 python tools/field_bench.py 3000000 10 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
  SDFX_FIELD_BWD_WAVES=$W timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "field" -p no:cacheprovider 2>&1 | tail -2 | tee -a $OUT/summary.txt
done
