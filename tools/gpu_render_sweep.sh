#!/bin/bash
# GPU-side durations (rocprofv3 kernel trace; tools/render_bench.py itself is bound by the Python launch path) of the fused render
# kernels for 1, 2, 4, 8 sibling waves per ray.
TAG=${1:-render_sweep}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
# the SDFX_* kernel switches exist only in the devtools library (include/sdfx_devtools.h)
export SDFX_LIB=${SDFX_LIB:-$PWD/stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so}
REPO=$PWD
cd /tmp
for W in 1 2 4 8; do
  SDFX_RENDER_WAVES=$W rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$W -o rb -- python $REPO/tools/render_bench.py > $OUT/rb$W.log 2>&1
  python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob
f = glob.glob("$OUT/p$W/**/rb_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_render_train" in r["Name"]:
        print("waves per ray $W  %-22s calls %s avg %.1f us min %.1f max %.1f" % (r["Name"].split("::")[1][:18], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
find $OUT -type f -size +1M -delete 2>/dev/null
