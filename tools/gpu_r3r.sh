#!/bin/bash
TAG=${1:-r3r}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for V in normal nocursor; do
  if [ $V = nocursor ]; then export SDFX_GRIDBWD_NOCURSOR=1; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$V -o t -- python $REPO/tools/gridbwd_bench.py 10 > $OUT/$V.log 2>&1
  echo "-- $V" | tee -a $OUT/summary.txt
  python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob
f = glob.glob("$OUT/$V/**/t_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_grid_bwd" in r["Name"]:
        print("%-40s calls %s avg %.1f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
find $OUT -type f -size +1M -delete 2>/dev/null
