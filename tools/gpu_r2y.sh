#!/bin/bash
TAG=${1:-r2y}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_sds.py tests/test_gpu_parity.py -m gpu -q --timeout 900 -p no:cacheprovider -k "stencil or sds or field" > $OUT/pytest.txt 2>&1; grep -E "^E  |passed|failed" $OUT/pytest.txt | cut -c1-700 | head -40 | tee -a $OUT/summary.txt
for V in 0 1; do
  for B in 3150000 1400000; do
    echo "-- SDFX_FIELD_BWD_LDSFRAG=$V B=$B" | tee -a $OUT/summary.txt
    SDFX_FIELD_BWD_LDSFRAG=$V timeout 300 python tools/field_bench.py $B 30 2>&1 | tail -1 | tee -a $OUT/summary.txt
  done
done
