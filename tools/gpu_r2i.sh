#!/bin/bash
TAG=${1:-r2i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_infer.py tests/test_gpu_occupancy.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -40 | cut -c1-260 | tee -a $OUT/summary.txt
timeout 600 python tools/infer_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real | tee -a $OUT/summary.txt
grep "bench +" $OUT/bench.err | tee -a $OUT/summary.txt
tail -3 $OUT/bench.err | cut -c1-300 | tee -a $OUT/summary.txt
