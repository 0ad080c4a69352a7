#!/bin/bash
TAG=${1:-r3e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 -p no:cacheprovider -k "row_limit or field or hinted" > $OUT/pytest.txt 2>&1; grep -E "^E  |passed|failed|^FAILED" $OUT/pytest.txt | cut -c1-500 | head -20 | tee -a $OUT/summary.txt
