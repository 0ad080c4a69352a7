#!/bin/bash
TAG=${1:-r2f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x 2>&1 | tail -40 > $OUT/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" | tee -a $OUT/summary.txt
tail -40 $OUT/pytest_gpu.log | cut -c1-300 | tee -a $OUT/summary.txt
for PH in latent rgb; do
  timeout 600 python bench.py --steps 30 --warmup 5 --phase $PH --guidance synthetic --no-cpu-baseline --no-kernel-bench > $OUT/bench_synth_$PH.json 2> $OUT/bench_synth_$PH.err
  echo "bench synthetic $PH exit: $?" | tee -a $OUT/summary.txt
  tail -3 $OUT/bench_synth_$PH.err | cut -c1-400 | tee -a $OUT/summary.txt
  python tools/pick_bench.py < $OUT/bench_synth_$PH.json 2>&1 | tee -a $OUT/summary.txt
done
