#!/bin/bash
TAG=${1:-r3l}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
PER_LEVEL=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o t -- python $REPO/tools/gridbwd_bench.py 3 > $OUT/log.txt 2>&1
python3 - <<PY | tee $OUT/summary.txt
import csv, glob
f = glob.glob("$OUT/prof/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "k_grid_bwd" in r["Kernel_Name"]]
bins = [d for n, d in seq if "bwd_bin" in n]
reds = [d for n, d in seq if "reduce" in n]
fins = [d for n, d in seq if "finish" in n]
print("launches: bin %d reduce %d finish %d" % (len(bins), len(reds), len(fins)))
# first 4 launches: full 16 levels; then 16 groups of 4 (max_level = 1..16)
print("full: bin %.1f reduce %.1f finish %.1f" % (bins[3], reds[3], fins[3] if len(fins) > 3 else 0))
prev = (0, 0)
for ml in range(1, 17):
    i = 4 + (ml - 1) * 4 + 3
    print("max_level=%2d  bin %7.1f (+%6.1f)  reduce %7.1f (+%6.1f)" % (ml, bins[i], bins[i] - prev[0], reds[i], reds[i] - prev[1]))
    prev = (bins[i], reds[i])
PY
find $OUT -type f -size +1M -delete 2>/dev/null
