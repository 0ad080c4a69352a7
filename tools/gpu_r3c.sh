#!/bin/bash
TAG=${1:-r3c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest.txt 2>&1; grep -E "^E  |passed|failed|^FAILED" $OUT/pytest.txt | cut -c1-400 | head -20 | tee -a $OUT/summary.txt
for V in 0 1; do
  for PH in mix; do
    echo "-- SDFX_ROW_LIMIT=$V $PH" | tee -a $OUT/summary.txt
    SDFX_ROW_LIMIT=$V timeout 600 python bench.py --steps 40 --warmup 8 --phase $PH --guidance synthetic --no-cpu-baseline --no-kernel-bench --no-reference-flow > $OUT/bench_synth_${PH}_$V.json 2> $OUT/bench_synth_${PH}_$V.err
    python tools/pick_bench.py < $OUT/bench_synth_${PH}_$V.json 2>&1 | cut -c1-300 | tee -a $OUT/summary.txt
  done
done
timeout 300 python tools/field_bench.py 3150000 30 2>&1 | tail -1 | tee -a $OUT/summary.txt
