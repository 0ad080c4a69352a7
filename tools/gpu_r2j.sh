#!/bin/bash
TAG=${1:-r2j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
for W in 2 3 4; do
  echo "SDFX_INFER_WAVES=$W" | tee -a $OUT/summary.txt
  SDFX_INFER_WAVES=$W HW=64,256,800 SCENES=blobs timeout 600 python tools/infer_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
done
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real | tee -a $OUT/summary.txt
grep "bench +" $OUT/bench.err | tee -a $OUT/summary.txt
tail -3 $OUT/bench.err | cut -c1-300 | tee -a $OUT/summary.txt
