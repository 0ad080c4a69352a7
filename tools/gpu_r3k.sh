#!/bin/bash
TAG=${1:-r3k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
PER_LEVEL=1 timeout 300 python tools/gridbwd_bench.py 5 2>&1 | tail -18 | tee -a $OUT/summary.txt
