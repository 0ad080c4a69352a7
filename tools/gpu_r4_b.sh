#!/bin/bash
# round 4, call B: product suite (second lease), the same suite on the devtools library, reference-kernel goldens for
# tv / wd / freq / SH, the default bench line, and a kernel-stats profile of the RGB phase.
TAG=${1:-r4b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q -rA --durations=10 --timeout 600 -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest (product) exit: $?" | tee -a $OUT/summary.txt
grep -E "^E  |passed|failed|^FAILED|^ERROR|^SKIPPED" $OUT/pytest.txt | cut -c1-300 | head -20 | tee -a $OUT/summary.txt
SDFX_LIB=$PWD/stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so timeout 900 python -m pytest tests -m gpu -q -rA --timeout 600 -p no:cacheprovider > $OUT/pytest_dev.txt 2>&1
echo "pytest (devtools) exit: $?" | tee -a $OUT/summary.txt
grep -E "^E  |passed|failed|^FAILED|^ERROR|^SKIPPED" $OUT/pytest_dev.txt | cut -c1-300 | head -20 | tee -a $OUT/summary.txt
timeout 600 python tests/golden/make_goldens_from_reference_kernels.py > $OUT/goldens.log 2>&1
echo "goldens exit: $?" | tee -a $OUT/summary.txt; tail -2 $OUT/goldens.log | cut -c1-400 | tee -a $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit: $?" | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
d=json.load(open("$OUT/bench.json"))
print({k:d.get(k) for k in ("value","ms_per_step","phases","iters_per_sec_nerf_only","iters_per_sec_without_unet","xcd_round_robin","ms_per_step_per_rank")})
print("roofline", {k:d["roofline"].get(k) for k in ("achieved","frac","avg_launch_us","points_per_launch","hbm_frac")})
print("cpu", d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o rgb -- python $OLDPWD/bench.py --phase rgb --steps 10 --warmup 4 --no-cpu-baseline --no-kernel-bench --no-nerf-only --no-reference-flow > $OLDPWD/$OUT/prof.log 2>&1 )
echo "rocprof exit: $?" | tee -a $OUT/summary.txt
find $OUT/prof -name "*kernel_stats*" | head -2 | tee -a $OUT/summary.txt
find $OUT/prof -type f -size +2M -delete 2>/dev/null
du -sh $OUT gpurun_out/golden | tee -a $OUT/summary.txt
