#!/bin/bash
TAG=${1:-r2z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest.txt 2>&1; grep -E "^E  |passed|failed|^FAILED" $OUT/pytest.txt | cut -c1-500 | head -30 | tee -a $OUT/summary.txt
for V in 0 1; do
  echo "-- SDFX_FIELD_BWD_LDSFRAG=$V" | tee -a $OUT/summary.txt
  SDFX_FIELD_BWD_LDSFRAG=$V timeout 300 python tools/field_bench.py 3150000 30 2>&1 | tail -1 | tee -a $OUT/summary.txt
done
