#!/bin/bash
# round 4, call H: suite on the final code, A/B of the last prior changes (1x1 convolutions as GEMMs + GEGLU kernel + hoisted SiLU
# = SDFX_BLOCK_FUSION, now wider), then the evidence: default line, IF / DMTet / synthetic lines, profile round (immediate-mode stats).
TAG=${1:-r4h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q -rA --durations=5 --timeout 600 -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest exit: $?" | tee -a $OUT/summary.txt
grep -E "^E  |passed|failed|^FAILED|^ERROR|Fatal" $OUT/pytest.txt | cut -c1-300 | head -12 | tee -a $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit: $?" | tee -a $OUT/summary.txt; tail -1 $OUT/smoke.log | cut -c1-300 | tee -a $OUT/summary.txt
Q="--steps 12 --warmup 4 --no-cpu-baseline --no-kernel-bench --no-nerf-only --no-reference-flow"
for PH in latent rgb; do
for V in "SDFX_BLOCK_FUSION=0" "SDFX_BLOCK_FUSION=1"; do
  SDFX_CONV_FIND=0 env $V timeout 900 python bench.py --phase $PH $Q > $OUT/bench_${PH}_$V.json 2> $OUT/bench_${PH}_$V.err
  echo "$PH $V (immediate-mode convs) exit $?: $(python -c "import json,sys; d=json.load(open('$OUT/bench_${PH}_$V.json')); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)" | tee -a $OUT/summary.txt
done; done
T0=$SECONDS
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench default exit: $? wall $((SECONDS-T0)) s" | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
d=json.load(open("$OUT/bench_default.json"))
print({k:d.get(k) for k in ("value","ms_per_step","phases","iters_per_sec_nerf_only","ms_nerf_only","iters_per_sec_without_unet","iters_per_sec_reference_flow","xcd_round_robin","scatter_overflow","seconds_to_first_barrier_per_rank")})
print("roofline", {k:d["roofline"].get(k) for k in ("achieved","frac","avg_launch_us","points_per_launch","hbm_frac")}, "gather", {k:d["roofline"].get("gather",{}).get(k) for k in ("ta_busy_frac","frac","lines_per_point")})
PY
for V in "--prior if" "--stage dmtet" "--guidance synthetic"; do
  N=$(echo $V | tr -d ' -')
  timeout 900 python bench.py $V --no-cpu-baseline --no-kernel-bench > $OUT/bench_$N.json 2> $OUT/bench_$N.err
  echo "bench $V exit $?: $(python -c "import json; d=json.load(open('$OUT/bench_$N.json')); print(d['value'], d['ms_per_step'], d.get('phases'), d.get('scatter_overflow'))" 2>&1 | tail -1 | cut -c1-400)" | tee -a $OUT/summary.txt
done
SDFX_CONV_FIND=0 bash tools/gpu_profile_round.sh $TAG/round > $OUT/profile_round.log 2>&1
echo "profile round exit: $?" | tee -a $OUT/summary.txt
grep -E "exit|copyBuffer" $OUT/profile_round.log | cut -c1-200 | tee -a $OUT/summary.txt
du -sh $OUT | tee -a $OUT/summary.txt
