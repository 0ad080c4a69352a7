#!/bin/bash
TAG=${1:-r2x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_sds.py -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest.txt 2>&1; grep -E "^E  |passed|failed" $OUT/pytest.txt | cut -c1-600 | head -40 | tee -a $OUT/summary.txt
timeout 600 python tools/zero_grad_rows.py latent 2>$OUT/zg.err | tail -9 | tee -a $OUT/summary.txt
for PH in rgb latent; do
  timeout 600 python bench.py --steps 40 --warmup 8 --phase $PH --guidance synthetic --no-cpu-baseline --no-kernel-bench --no-reference-flow > $OUT/bench_synth_$PH.json 2> $OUT/bench_synth_$PH.err
  python tools/pick_bench.py < $OUT/bench_synth_$PH.json 2>&1 | tee -a $OUT/summary.txt
  python -c "
import json,sys
for l in open('$OUT/bench_synth_$PH.json'):
    if l.startswith('{'):
        d=json.loads(l); print(d['phases'], d['graph_stats_timed_region'])
" | tee -a $OUT/summary.txt
done
tail -3 $OUT/zg.err
