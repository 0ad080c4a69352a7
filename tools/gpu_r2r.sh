#!/bin/bash
TAG=${1:-r2r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_sds.py tests/test_gpu_trainer.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -25 | cut -c1-250 | tee -a $OUT/summary.txt
for PH in latent rgb; do
  timeout 600 python bench.py --steps 40 --warmup 8 --phase $PH --guidance synthetic --no-cpu-baseline --no-kernel-bench --no-reference-flow > $OUT/bench_synth_$PH.json 2> $OUT/bench_synth_$PH.err
  python tools/pick_bench.py < $OUT/bench_synth_$PH.json 2>&1 | tee -a $OUT/summary.txt
  tail -2 $OUT/bench_synth_$PH.err | cut -c1-300 | tee -a $OUT/summary.txt
done
bash tools/gpu_iter_trace.sh $TAG/lat latent 2>&1 | tail -4 | tee -a $OUT/summary.txt
bash tools/gpu_iter_trace.sh $TAG/rgb rgb 2>&1 | tail -4 | tee -a $OUT/summary.txt
