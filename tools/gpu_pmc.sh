#!/bin/bash
# PMC passes over the encode-forward launch loop (separate passes; no tracing flags besides kernel-trace).
TAG=${1:-pmc}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TCC|SQ|TA|TD|GRBM)_[A-Za-z0-9_]+" | sort -u > $OUT/counters.txt
wc -l $OUT/counters.txt
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for KIND in ${KINDS:-uniform ray}; do
    timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p${i}_$KIND -o pmc -- python $REPO/tools/encode_bench.py $KIND f16 3 > $OUT/p${i}_$KIND.log 2>&1
    echo "set $i ($SET) $KIND exit $?" | tee -a $OUT/summary.txt
    tail -1 $OUT/p${i}_$KIND.log | tee -a $OUT/summary.txt
  done
done
find $OUT -type f -size +1M -delete 2>/dev/null
python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*_*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_grid_forward" in r.get("Kernel_Name", ""):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[-2], {k: (sum(v[-3:]) / max(len(v[-3:]), 1)) for k, v in agg.items()})
PY
du -sh $OUT
