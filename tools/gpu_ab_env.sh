#!/bin/bash
# A/B of one environment switch on (a) the standalone scatter loop, (b) the synthetic-prior iteration, with kernel stats of (b).
# Usage: bash tools/gpu_ab_env.sh <tag> <ENV_NAME> <value A> <value B> [tests to run first]
TAG=${1:-ab}; VAR=$2; A=$3; B=$4; TESTS=${5:-}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
# the SDFX_* kernel switches exist only in the devtools library (include/sdfx_devtools.h)
export SDFX_LIB=${SDFX_LIB:-$PWD/stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so}
REPO=$PWD
if [ -n "$TESTS" ]; then
  python -m pytest $TESTS -q -m gpu --no-header -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-600 | tee $OUT/tests.log
fi
for V in $A $B; do
  echo "== $VAR=$V" | tee -a $OUT/summary.txt
  env $VAR=$V python tools/gridbwd_bench.py 20 2>&1 | tail -1 | tee -a $OUT/summary.txt
  env $VAR=$V python bench.py --steps 40 --warmup 8 --guidance synthetic --no-cpu-baseline --no-kernel-bench --no-reference-flow > $OUT/bench_$V.json 2> $OUT/bench_$V.err
  python tools/pick_bench.py < $OUT/bench_$V.json 2>&1 | cut -c1-300 | tee -a $OUT/summary.txt
  ( cd /tmp && env $VAR=$V timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$V -o bench -- python $REPO/bench.py --steps 40 --warmup 8 --guidance synthetic --no-cpu-baseline --no-kernel-bench --no-reference-flow --no-nerf-only > $OUT/prof_$V.log 2>&1 )
  f=$(find $OUT/prof_$V -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY' | tee -a $OUT/summary.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("  %-58s calls %5s avg %9.1f us %6s%%" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:58], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
  cp "$f" $OUT/kernel_stats_$V.csv
  find $OUT/prof_$V -type f -size +1M -delete
done
