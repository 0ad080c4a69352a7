#!/bin/bash
# bench's replayed-graph figure of the encode against rocprofv3's per-kernel average of the same command (synthetic prior)
TAG=${1:-rpl}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
CMD="python $REPO/bench.py --steps 40 --warmup 8 --guidance synthetic --no-cpu-baseline --no-kernel-bench --no-reference-flow --no-children"
$CMD > $OUT/bench.json 2> $OUT/bench.err
python tools/pick_bench.py < $OUT/bench.json 2>&1 | cut -c1-400 | tee $OUT/summary.txt
tail -3 $OUT/bench.err | cut -c1-300
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- $CMD > $OUT/prof.json 2> $OUT/prof.log )
python tools/pick_bench.py < $OUT/prof.json 2>&1 | grep roofline | cut -c1-300 | tee -a $OUT/summary.txt
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee -a $OUT/summary.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("  %-58s calls %5s avg %9.1f us %6s%%" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:58], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cp "$f" $OUT/kernel_stats.csv; find $OUT/prof -type f -size +1M -delete
