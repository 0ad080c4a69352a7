#!/bin/bash
# round 4, evidence on the final code of the later sessions (prior kernels: conv.hip, attention.hip, one-launch GroupNorm): the whole GPU
# suite + smoke, the four bench configurations, rocprofv3 kernel stats of the default command. (The PMC passes of tools/gpu_profile_round.sh
# cover the NeRF kernels, which these sessions did not touch: profiles/r04_pmc_traffic.json stands.)
TAG=${1:-r4final2}
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
print({k:d.get(k) for k in ("value","ms_per_step","phases","iters_per_sec_nerf_only","ms_nerf_only","iters_per_sec_without_unet","iters_per_sec_reference_flow","xcd_round_robin")})
print("roofline", {k:d["roofline"].get(k) for k in ("achieved","frac","avg_launch_us","points_per_launch","hbm_frac")})
print("cpu", d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
for V in "--prior if" "--stage dmtet" "--guidance synthetic"; do
  N=$(echo $V | tr -d ' -')
  timeout 600 python bench.py $V --no-cpu-baseline --no-kernel-bench > $OUT/bench_$N.json 2> $OUT/bench_$N.err
  echo "bench $V exit $?: $(python -c "import json; d=json.load(open('$OUT/bench_$N.json')); print(d['value'], d['ms_per_step'], d.get('phases'))" 2>&1 | tail -1 | cut -c1-300)" | tee -a $OUT/summary.txt
done
REPO=$PWD
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/stats -o bench -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-bench --no-reference-flow --no-nerf-only > $REPO/$OUT/stats.log 2>&1
echo "stats exit $?" | tee -a $REPO/$OUT/summary.txt
cd $REPO
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
rm -rf $OUT/stats
head -30 $OUT/bench_kernel_stats.csv | cut -c1-150 | tee -a $OUT/summary.txt
