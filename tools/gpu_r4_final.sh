#!/bin/bash
# round 4, evidence run on the final code: suite + smoke, the four bench configurations, standalone launch loops, kernel stats and
# PMC passes (traffic + gather counters) reduced to pmc_traffic.json. Copy the summaries into profiles/r04_*.
TAG=${1:-r4final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
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
print({k:d.get(k) for k in ("value","ms_per_step","phases","iters_per_sec_nerf_only","ms_nerf_only","iters_per_sec_without_unet","iters_per_sec_reference_flow","xcd_round_robin","seconds_to_first_barrier_per_rank")})
print("roofline", {k:d["roofline"].get(k) for k in ("achieved","frac","avg_launch_us","points_per_launch","hbm_frac","gather")})
print("composite", d.get("roofline_composite",{}).get("forward",{}).get("avg_launch_us"), d.get("roofline_composite",{}).get("backward",{}).get("avg_launch_us"))
print({k:v for k,v in d.get("kernels_standalone",{}).items() if k.startswith("composite")})
print("cpu", d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
for V in "--prior if" "--stage dmtet" "--guidance synthetic"; do
  N=$(echo $V | tr -d ' -')
  timeout 900 python bench.py $V --no-cpu-baseline --no-kernel-bench > $OUT/bench_$N.json 2> $OUT/bench_$N.err
  echo "bench $V exit $?: $(python -c "import json; d=json.load(open('$OUT/bench_$N.json')); print(d['value'], d['ms_per_step'], d.get('phases'))" 2>&1 | tail -1 | cut -c1-300)" | tee -a $OUT/summary.txt
done
timeout 200 python tools/gridbwd_bench.py 20 2>&1 | tail -2 | tee -a $OUT/summary.txt
SDFX_LIB=$PWD/stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so timeout 300 python tools/encode_bench.py stencil f16 10 2>&1 | tail -4 | cut -c1-400 | tee -a $OUT/summary.txt
bash tools/gpu_profile_round.sh $TAG/round > $OUT/profile_round.log 2>&1
echo "profile round exit: $?" | tee -a $OUT/summary.txt
tail -25 $OUT/profile_round.log | cut -c1-600 | tee -a $OUT/summary.txt
du -sh $OUT | tee -a $OUT/summary.txt
