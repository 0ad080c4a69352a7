#!/bin/bash
# A/B of a PYTHON-side switch of the package (an SDFX_* variable read by _devswitch in a devtools session) by rocprofv3 kernel stats of the
# synthetic-prior bench: that run is bit-reproducible, so both variants replay the same sequence of sample counts and their per-kernel
# averages compare directly. Two interleaved rounds.   gpurun -- 'bash tools/gpu_ab_pyswitch.sh <tag> SDFX_BASE_ALBEDO 0 1 [kernel name filter]'
TAG=$1; VAR=$2; A=$3; B=$4;   # VAR may be several names joined by commas: all set to the same value
 FILT=${5:-"k_field_|k_grid_|k_render|k_adan|Fill|copyBuffer"}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp SDFX_DEV=1
REPO=$PWD
cd /tmp
for rnd in 1 2; do
  for V in $A $B; do
    env $(echo $VAR | tr ',' '\n' | sed "s/$/=$V/" | tr '\n' ' ') timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p${V}_$rnd -o bench -- python $REPO/bench.py --steps 20 --warmup 3 --guidance synthetic --no-cpu-baseline --no-kernel-bench --no-reference-flow --no-nerf-only --no-children > $OUT/p${V}_$rnd.log 2>&1
    python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob, re
f = glob.glob("$OUT/p${V}_$rnd/**/bench_kernel_stats.csv", recursive=True)[0]
tot = 0.0
print("== $VAR=$V round $rnd")
for r in csv.DictReader(open(f)):
    tot += float(r["TotalDurationNs"])
    if re.search(r"$FILT", r["Name"]):
        print("  %-46s calls %5s avg %8.1f us total %8.2f ms" % (r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
print("  all kernels: %.2f ms" % (tot / 1e6))
PY
  done
done
find $OUT -type f -size +1M -delete 2>/dev/null
