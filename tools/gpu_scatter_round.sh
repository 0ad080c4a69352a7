#!/bin/bash
# Scatter (K1/K2/K3) check: parity tests, standalone loop per level, instruction counters of K1 / K2.
TAG=${1:-scatter}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
python -m pytest tests/test_gpu_02_parity.py tests/test_gpu_00_vs_reference_kernels.py tests/test_gpu_01_reference_goldens.py tests/test_gpu_zz_stress.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tee $OUT/tests.log
PER_LEVEL=${PER_LEVEL:-} python tools/gridbwd_bench.py 20 2>&1 | tail -18 | tee $OUT/bench.txt
cd /tmp
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p1 -o pmc -- python $REPO/tools/gridbwd_bench.py 1 > $OUT/p1.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -o st -- python $REPO/tools/gridbwd_bench.py 10 > $OUT/st.log 2>&1
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    for kn in ("k_grid_bwd_bin", "k_grid_bwd_reduce"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kn in r.get("Kernel_Name", ""):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        d = {k: v[-1] for k, v in agg.items()}
        print(kn, d)
        if d.get("SQ_WAVES"):
            print("   per wave: VALU %.0f SALU %.0f LDS %.1f" % (d["SQ_INSTS_VALU"] / d["SQ_WAVES"], d["SQ_INSTS_SALU"] / d["SQ_WAVES"], d["SQ_INSTS_LDS"] / d["SQ_WAVES"]))
for f in glob.glob("$OUT/st/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:5]:
        print("  %-50s calls %5s avg %9.1f us" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find $OUT -type f -size +1M -delete 2>/dev/null
