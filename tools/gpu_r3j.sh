#!/bin/bash
TAG=${1:-r3j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 -p no:cacheprovider -k "row_limit or field" > $OUT/pytest.txt 2>&1; grep -E "^E  |passed|failed|^FAILED" $OUT/pytest.txt | cut -c1-500 | head -20 | tee -a $OUT/summary.txt
timeout 300 python tools/field_bench.py 3150000 30 2>&1 | tail -1 | tee -a $OUT/summary.txt
bash tools/gpu_pmc_gridbwd.sh $TAG/pmc 2>&1 | tail -8 | cut -c1-900 | tee -a $OUT/summary.txt
