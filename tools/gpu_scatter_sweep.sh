#!/bin/bash
# Scatter parameter sweep on the standalone loop (kernel-trace durations): bucket size (variant library), run-folding threshold.
TAG=${1:-scatter_sweep}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
# the SDFX_* kernel switches exist only in the devtools library (include/sdfx_devtools.h)
export SDFX_LIB=${SDFX_LIB:-$PWD/stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so}
REPO=$PWD
python -m pytest tests/test_gpu_02_parity.py tests/test_gpu_zz_stress.py -q -m gpu --no-header -p no:cacheprovider -k "stencil_source or grid_backward or binned" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-600 | tee $OUT/tests.log
SDFX_LIB=$REPO/stable-dreamfusion_amd/csrc/libsdfx_hip_b12.so python -m pytest tests/test_gpu_02_parity.py tests/test_gpu_zz_stress.py -q -m gpu --no-header -p no:cacheprovider -k "grid_backward or binned" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-300 | tee -a $OUT/tests.log
cd /tmp
run() {  # label, env...
  L=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$L -o st -- python $REPO/tools/gridbwd_bench.py 10 > $OUT/$L.log 2>&1
  python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob
f = glob.glob("$OUT/p_$L/**/st_kernel_stats.csv", recursive=True)[0]
d = {r["Name"].split("::")[1].split("(")[0].split("<")[0]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f)) if "k_grid_bwd" in r["Name"]}
print("%-28s" % "$L", "  ".join("%s %.1f" % (k.replace("k_grid_bwd_", ""), v) for k, v in sorted(d.items())), " sum %.1f us" % sum(d.values()))
PY
}
run bucket2048 SDFX_X=0
run bucket4096 SDFX_LIB=$REPO/stable-dreamfusion_amd/csrc/libsdfx_hip_b12.so
for M in 0 160 300 420 800 1023; do run merge_res_$M SDFX_GRIDBWD_MERGE_RES=$M; done
run b4096_merge_res_420 SDFX_LIB=$REPO/stable-dreamfusion_amd/csrc/libsdfx_hip_b12.so SDFX_GRIDBWD_MERGE_RES=420
find $OUT -type f -size +1M -delete 2>/dev/null
