#!/bin/bash
# round 4, on the final commit (capture-safe caches, shared xcd_contiguous): the whole GPU suite, smoke, the default bench (same kernels as tools/gpu_r4_final2.sh's run)
OUT=gpurun_out/r4final4
mkdir -p $OUT
echo "== $(date)" | tee $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q -rA --durations=5 --timeout 600 -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest exit: $?" | tee -a $OUT/summary.txt
grep -E "^E  |passed|failed|^FAILED|^ERROR|Fatal" $OUT/pytest.txt | cut -c1-300 | head -12 | tee -a $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit: $?" | tee -a $OUT/summary.txt; tail -1 $OUT/smoke.log | cut -c1-300 | tee -a $OUT/summary.txt
T0=$SECONDS
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench default exit: $? wall $((SECONDS-T0)) s" | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
d=json.load(open("$OUT/bench_default.json"))
print({k:d.get(k) for k in ("value","ms_per_step","phases","iters_per_sec_nerf_only","ms_nerf_only","iters_per_sec_without_unet","iters_per_sec_reference_flow","xcd_round_robin")})
print("roofline", {k:d["roofline"].get(k) for k in ("achieved","frac","avg_launch_us","points_per_launch","hbm_frac")})
PY
