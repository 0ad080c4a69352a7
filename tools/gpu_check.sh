#!/bin/bash
# One gpurun call: parity tests, smoke, a short bench, and a rocprofv3 kernel-stats pass.
# Usage (from the repo root on the GPU box): bash tools/gpu_check.sh [tag]
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
rocm-smi --showproductname 2>/dev/null | head -8 >> $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -60 > $OUT/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" | tee -a $OUT/summary.txt
tail -15 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit: $?" | tee -a $OUT/summary.txt
tail -5 $OUT/smoke.log | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit: $?" | tee -a $OUT/summary.txt
tail -3 $OUT/bench.err | tee -a $OUT/summary.txt
cat $OUT/bench.json | tee -a $OUT/summary.txt
if [ "${PROFILE:-1}" = "1" ]; then
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o bench -- python $OLDPWD/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-bench > $OLDPWD/$OUT/prof.log 2>&1 )
  echo "rocprof exit: $?" | tee -a $OUT/summary.txt
  find $OUT/prof -name "*kernel_stats*" | head -3 | tee -a $OUT/summary.txt
  # keep the small summaries, drop everything big (gpurun_out is capped at 64 MiB)
  find $OUT/prof -type f -size +2M -delete 2>/dev/null
  du -sh $OUT | tee -a $OUT/summary.txt
fi
