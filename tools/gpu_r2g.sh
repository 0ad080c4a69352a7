#!/bin/bash
TAG=${1:-r2g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real | tee -a $OUT/summary.txt
echo "bench exit: $?" | tee -a $OUT/summary.txt
tail -5 $OUT/bench.err | cut -c1-400 | tee -a $OUT/summary.txt
cat $OUT/bench.json | cut -c1-6000 | tee -a $OUT/summary.txt
