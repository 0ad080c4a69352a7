#!/bin/bash
# GPU-side durations of the fused render kernels against the sample count (tools/render_scaling.py), rocprofv3 kernel stats per run.
TAG=${1:-render_scaling}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for F in 0.1 0.2 0.25 0.35 0.5 1.0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$F -o rs -- python $REPO/tools/render_scaling.py $F > $OUT/rs$F.log 2>&1
  tail -1 $OUT/rs$F.log | tee -a $OUT/summary.txt
  python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob
f = glob.glob("$OUT/p$F/**/rs_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_render_train" in r["Name"]:
        print("   %-22s calls %s avg %.1f us min %.1f max %.1f" % (r["Name"].split("::")[1][:18], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
find $OUT -type f -size +1M -delete 2>/dev/null
