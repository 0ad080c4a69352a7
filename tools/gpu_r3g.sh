#!/bin/bash
TAG=${1:-r3g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest.txt 2>&1; grep -E "^E  |passed|failed|^FAILED" $OUT/pytest.txt | cut -c1-400 | head -20 | tee -a $OUT/summary.txt
for V in 0 1; do
  echo "-- SDFX_FIELD_BWD_NAT=$V mix" | tee -a $OUT/summary.txt
  SDFX_FIELD_BWD_NAT=$V timeout 600 python bench.py --steps 40 --warmup 8 --phase mix --guidance synthetic --no-cpu-baseline --no-kernel-bench --no-reference-flow > $OUT/bench_synth_mix_$V.json 2> $OUT/bench_synth_mix_$V.err
  python tools/pick_bench.py < $OUT/bench_synth_mix_$V.json 2>&1 | cut -c1-300 | tee -a $OUT/summary.txt
done
