#!/bin/bash
TAG=${1:-pmcf}
ROWS=${2:-420000}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
python $REPO/tools/field_bench.py $ROWS 5 2>&1 | tail -1 | tee $OUT/summary.txt
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQC_ICACHE_MISSES SQC_ICACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_HITS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o pmc -- python $REPO/tools/field_bench.py $ROWS 2 > $OUT/p$i.log 2>&1
  echo "set $i exit $?" | tee -a $OUT/summary.txt
done
find $OUT -type f -size +1M -delete 2>/dev/null
python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    for kn in ("k_field_backward", "k_field_forward"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kn in r.get("Kernel_Name", ""):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(kn, {k: v[-1] for k, v in agg.items()})
PY
