#!/bin/bash
# round 4, call C: GroupNorm kernels — their tests, the whole suite, and the A/B of both phases (SDFX_GROUPNORM=0 / 1).
TAG=${1:-r4c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_04_sds.py tests/test_gpu_00_vs_reference_kernels.py -m gpu -q -x --timeout 300 -p no:cacheprovider > $OUT/pytest_gn.txt 2>&1
echo "pytest (groupnorm, goldens) exit: $?" | tee -a $OUT/summary.txt
grep -E "^E  |passed|failed|^FAILED|^ERROR" $OUT/pytest_gn.txt | cut -c1-400 | head -30 | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest (all) exit: $?" | tee -a $OUT/summary.txt
grep -E "^E  |passed|failed|^FAILED|^ERROR" $OUT/pytest.txt | cut -c1-300 | head -20 | tee -a $OUT/summary.txt
Q="--steps 12 --warmup 4 --no-cpu-baseline --no-kernel-bench --no-nerf-only --no-reference-flow"
for PH in rgb latent; do
for V in "SDFX_GROUPNORM=0" "SDFX_GROUPNORM=1"; do
  env $V timeout 400 python bench.py --phase $PH $Q > $OUT/bench_${PH}_$V.json 2> $OUT/bench_${PH}_$V.err
  echo "$PH $V exit $?: $(python -c "import json,sys; d=json.load(open('$OUT/bench_${PH}_$V.json')); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)" | tee -a $OUT/summary.txt
done; done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o rgb -- python $OLDPWD/bench.py --phase rgb $Q > $OLDPWD/$OUT/prof.log 2>&1 )
echo "rocprof exit: $?" | tee -a $OUT/summary.txt
find $OUT/prof -type f -size +2M -delete 2>/dev/null
du -sh $OUT | tee -a $OUT/summary.txt
