#!/bin/bash
TAG=${1:-r3a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 300 python tools/overlap_bench.py 2>$OUT/ov.err | tail -2 | tee -a $OUT/summary.txt
tail -3 $OUT/ov.err
