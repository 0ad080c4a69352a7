#!/bin/bash
# rocprofv3 kernel trace of the bench in a given train mode; keeps stats + a per-step gap analysis
TAG=${1:-pg}; MODE=${2:-graph}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
SDFX_DEV=1 SDFX_BENCH_TRACE=1 SDFX_TRAIN_MODE=$MODE timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-bench > $OUT/prof.log 2>&1
echo "rocprof exit $?"
grep -E "trace|value" $OUT/prof.log | cut -c1-160
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/prof/bench_kernel_trace.csv")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith('k_march_count')]
print("kernels",len(rows),"steps",len(idx))
for a,b in list(zip(idx[:-1],idx[1:]))[-12:]:
    t0=int(rows[a]['Start_Timestamp']); t1=int(rows[b]['Start_Timestamp'])
    busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows[a:b])
    top=sorted(rows[a:b],key=lambda r:int(r['Start_Timestamp'])-int(r['End_Timestamp']))[:3]
    print("span %.2f ms busy %.2f ms n=%d top: %s"%((t1-t0)/1e6,busy/1e6,b-a," | ".join("%s %.0fus"%(r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','')[:24],(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in top)))
PY
find $OUT/prof -type f -size +2M -delete 2>/dev/null
