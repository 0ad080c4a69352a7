#!/bin/bash
# round 4, call D: suites (product + devtools) on the GroupNorm / top-left / devtools code, the GroupNorm A/B of both phases, and a
# kernel trace of the RGB phase reduced to: kernel stats + how busy the GPU is between two optimiser steps (is the step launch-bound?).
TAG=${1:-r4d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q -rA --durations=8 --timeout 600 -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest (product) exit: $?" | tee -a $OUT/summary.txt
grep -E "^E  |passed|failed|^FAILED|^ERROR|Fatal" $OUT/pytest.txt | cut -c1-300 | head -20 | tee -a $OUT/summary.txt
SDFX_LIB=$PWD/stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_dev.txt 2>&1
echo "pytest (devtools) exit: $?" | tee -a $OUT/summary.txt
grep -E "^E  |passed|failed|^FAILED|^ERROR|Fatal" $OUT/pytest_dev.txt | cut -c1-300 | head -20 | tee -a $OUT/summary.txt
Q="--steps 12 --warmup 4 --no-cpu-baseline --no-kernel-bench --no-nerf-only --no-reference-flow"
for PH in rgb latent; do
for V in "SDFX_GROUPNORM=0" "SDFX_GROUPNORM=1"; do
  env $V timeout 400 python bench.py --phase $PH $Q > $OUT/bench_${PH}_$V.json 2> $OUT/bench_${PH}_$V.err
  echo "$PH $V exit $?: $(python -c "import json,sys; d=json.load(open('$OUT/bench_${PH}_$V.json')); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)" | tee -a $OUT/summary.txt
done; done
for PH in rgb latent; do
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_$PH -o $PH -- python $OLDPWD/bench.py --phase $PH $Q > $OLDPWD/$OUT/prof_$PH.log 2>&1 )
echo "rocprof $PH exit: $?" | tee -a $OUT/summary.txt
python3 - <<PY | tee -a $OUT/summary.txt
import csv
rows = list(csv.DictReader(open("$OUT/prof_$PH/${PH}_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps = [i for i, r in enumerate(rows) if "k_adan_update" in r["Kernel_Name"]]
# the last 8 optimiser steps of the timed region: wall span between their first and last k_adan_update, and the kernel time inside
if len(steps) >= 18:
    a, b = steps[-9 - 8], steps[-9]      # (the 8 eager roofline iterations come after the timed region: skip them)
    t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a + 1:b + 1])
    n = b - a
    print("$PH: 8 steps: wall %.2f ms/step, kernel time %.2f ms/step (busy %.1f %%), %d launches/step, mean kernel %.1f us"
          % ((t1 - t0) / 8e6, busy / 8e6, 100.0 * busy / (t1 - t0), n / 8, busy / n / 1e3))
PY
cp $OUT/prof_$PH/${PH}_kernel_stats.csv $OUT/${PH}_kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof_$PH
done
du -sh $OUT | tee -a $OUT/summary.txt
