#!/bin/bash
TAG=${1:-r3s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
for V in 0 1; do echo "-- SDFX_GRIDBWD_BALANCE=$V" | tee -a $OUT/summary.txt; SDFX_GRIDBWD_BALANCE=$V timeout 200 python tools/gridbwd_bench.py 15 2>&1 | tail -1 | tee -a $OUT/summary.txt; done
SDFX_GRIDBWD_BALANCE=1 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -m gpu -q --timeout 300 -p no:cacheprovider -k "grid or binned or reproduc or row_limit" > $OUT/pytest.txt 2>&1; grep -E "^E  |passed|failed|^FAILED" $OUT/pytest.txt | cut -c1-300 | head -8 | tee -a $OUT/summary.txt
