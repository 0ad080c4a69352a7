#!/bin/bash
TAG=${1:-r2l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -40 | cut -c1-260 > $OUT/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" | tee -a $OUT/summary.txt
tail -12 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit: $?" | tee -a $OUT/summary.txt
tail -3 $OUT/smoke.log | cut -c1-300 | tee -a $OUT/summary.txt
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real | tee -a $OUT/summary.txt
grep "bench +" $OUT/bench.err | tee -a $OUT/summary.txt
tail -2 $OUT/bench.err | cut -c1-300 | tee -a $OUT/summary.txt
bash tools/gpu_profile_round.sh ${TAG}_prof 2>&1 | tail -30 | cut -c1-400 | tee -a $OUT/summary.txt
