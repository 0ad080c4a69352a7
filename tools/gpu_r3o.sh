#!/bin/bash
TAG=${1:-r3o}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -m gpu -q --timeout 900 -p no:cacheprovider -k "grid or binned or reproduc or row_limit" > $OUT/pytest.txt 2>&1; grep -E "^E  |passed|failed|^FAILED" $OUT/pytest.txt | cut -c1-500 | head -20 | tee -a $OUT/summary.txt
for V in 0 1; do echo "-- SDFX_GRIDBWD_QUEUE=$V" | tee -a $OUT/summary.txt; SDFX_GRIDBWD_QUEUE=$V timeout 300 python tools/gridbwd_bench.py 20 2>&1 | tail -1 | tee -a $OUT/summary.txt; done
