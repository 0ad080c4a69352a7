#!/bin/bash
TAG=${1:-r2g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_occupancy.py tests/test_gpu_render.py tests/test_gpu_trainer.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -30 | cut -c1-250 | tee -a $OUT/summary.txt
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real | tee -a $OUT/summary.txt
tail -5 $OUT/bench.err | cut -c1-400 | tee -a $OUT/summary.txt
cat $OUT/bench.json | cut -c1-3000 | tee -a $OUT/summary.txt
