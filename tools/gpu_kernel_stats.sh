#!/bin/bash
# rocprofv3 kernel stats (top N kernels) of a bench.py invocation.  Usage: bash tools/gpu_kernel_stats.sh <tag> <top N> -- <bench args...>
TAG=${1:-kstats}; TOP=${2:-25}; shift 3
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
python3 - "$f" $TOP <<'PY' | tee $OUT/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU kernel time %.1f ms over %d kernel names" % (tot / 1e6, len(rows)))
for r in rows[:int(sys.argv[2])]:
    print("  %-64s calls %6s avg %9.1f us %6s%%" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
find $OUT -type f -size +1M -delete 2>/dev/null
