#!/bin/bash
# End-of-round evidence run: GPU tests, default bench line, synthetic-prior benches per phase, step timeline, iteration trace,
# rocprof kernel stats + PMC traffic (tools/gpu_profile_round.sh).  Usage: bash tools/gpu_evidence.sh <tag>
TAG=${1:-evidence}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest.txt 2>&1; grep -E "^E  |passed|failed|^FAILED" $OUT/pytest.txt | cut -c1-400 | head -20 | tee -a $OUT/summary.txt
( time timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real | tee -a $OUT/summary.txt
python tools/pick_bench.py < $OUT/bench_default.json 2>&1 | cut -c1-400 | tee -a $OUT/summary.txt
for PH in ${PHASES:-latent rgb mix}; do
  timeout 600 python bench.py --steps 40 --warmup 8 --phase $PH --guidance synthetic --no-cpu-baseline --no-kernel-bench --no-reference-flow > $OUT/bench_synth_$PH.json 2> $OUT/bench_synth_$PH.err
  python tools/pick_bench.py < $OUT/bench_synth_$PH.json 2>&1 | cut -c1-400 | tee -a $OUT/summary.txt
done
timeout 600 python tools/step_timeline.py latent 2>$OUT/tl.err | tail -7 | cut -c1-130 | tee $OUT/step_timeline_latent.txt | tee -a $OUT/summary.txt
bash tools/gpu_iter_trace.sh $TAG/lat latent 2>&1 | tail -22 | cut -c1-160 | tee -a $OUT/summary.txt
bash tools/gpu_profile_round.sh $TAG/prof 2>&1 | tail -12 | cut -c1-300 | tee -a $OUT/summary.txt
