#!/bin/bash
TAG=${1:-r3h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_render.py -m gpu -q --timeout 900 -p no:cacheprovider -k "row_limit or field or fused or render" > $OUT/pytest.txt 2>&1; grep -E "^E  |passed|failed|^FAILED" $OUT/pytest.txt | cut -c1-500 | head -20 | tee -a $OUT/summary.txt
run() { echo "-- $*" | tee -a $OUT/summary.txt; env "$@" timeout 300 python tools/field_bench.py 3150000 30 2>&1 | tail -1 | tee -a $OUT/summary.txt; }
run SDFX_FIELD_FWD_NAT=0
run SDFX_FIELD_FWD_NAT=1
run SDFX_FIELD_FWD_NAT=2
