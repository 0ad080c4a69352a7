#!/bin/bash
# round 4, call F: sds / GroupNorm tests, then A/Bs on both phases: block fusion (bias / time-embedding / residual folded), MIOpen find mode
TAG=${1:-r4f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_04_sds.py tests/test_gpu_05_trainer.py -m gpu -q --timeout 300 -p no:cacheprovider > $OUT/pytest_gn.txt 2>&1
echo "pytest (sds, trainer) exit: $?" | tee -a $OUT/summary.txt
grep -E "^E  |passed|failed|^FAILED|^ERROR" $OUT/pytest_gn.txt | cut -c1-300 | head -10 | tee -a $OUT/summary.txt
Q="--steps 12 --warmup 4 --no-cpu-baseline --no-kernel-bench --no-nerf-only --no-reference-flow"
for PH in rgb latent; do
for V in "SDFX_BLOCK_FUSION=0" "SDFX_BLOCK_FUSION=1" "SDFX_CONV_FIND=1"; do
  T0=$SECONDS
  env $V timeout 900 python bench.py --phase $PH $Q > $OUT/bench_${PH}_$V.json 2> $OUT/bench_${PH}_$V.err
  echo "$PH $V exit $?: $(python -c "import json,sys; d=json.load(open('$OUT/bench_${PH}_$V.json')); print(d['value'], d['ms_per_step'], d['seconds_to_first_barrier_per_rank'])" 2>&1 | tail -1) wall $((SECONDS-T0)) s" | tee -a $OUT/summary.txt
done; done
du -sh $OUT | tee -a $OUT/summary.txt
