#!/bin/bash
TAG=${1:-r2w}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -15 | cut -c1-250 | tee -a $OUT/summary.txt
for PH in latent rgb; do
  timeout 600 python bench.py --steps 40 --warmup 8 --phase $PH --guidance synthetic --no-cpu-baseline --no-kernel-bench --no-reference-flow > $OUT/bench_synth_$PH.json 2> $OUT/bench_synth_$PH.err
  python tools/pick_bench.py < $OUT/bench_synth_$PH.json 2>&1 | tee -a $OUT/summary.txt
done
bash tools/gpu_iter_trace.sh $TAG/lat latent 2>&1 | tail -24 | cut -c1-200 | tee -a $OUT/summary.txt
