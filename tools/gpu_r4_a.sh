#!/bin/bash
# round 4, call A: the whole GPU suite WITHOUT -x (per-test outcomes + durations), smoke, the scatter's launch loop, and the
# VAE-outside-autocast A/B of the RGB phase.
TAG=${1:-r4a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
rocm-smi --showproductname 2>/dev/null | grep -i "card series\|gfx" | head -2 >> $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=25 --timeout 600 -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest exit: $?" | tee -a $OUT/summary.txt
grep -E "^E  |passed|failed|^FAILED|^ERROR" $OUT/pytest.txt | cut -c1-400 | head -40 | tee -a $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit: $?" | tee -a $OUT/summary.txt
tail -3 $OUT/smoke.log | cut -c1-300 | tee -a $OUT/summary.txt
timeout 200 python tools/gridbwd_bench.py 20 2>&1 | tail -3 | tee -a $OUT/summary.txt
Q="--phase rgb --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-bench --no-nerf-only --no-reference-flow"
for V in "SDFX_VAE_AUTOCAST=1" "SDFX_VAE_AUTOCAST=0" "SDFX_VAE_CL=1"; do
  env $V timeout 400 python bench.py $Q > $OUT/bench_$V.json 2> $OUT/bench_$V.err
  echo "$V exit $?: $(python -c "import json,sys; d=json.load(open('$OUT/bench_$V.json')); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)" | tee -a $OUT/summary.txt
done
du -sh $OUT | tee -a $OUT/summary.txt
