#!/bin/bash
# The write-pattern microbenchmark (tools/ubench/write_streams.hip) under rocprofv3 --pmc: L2 -> fabric write requests by size
# (64-byte against partial), write-backs and read requests per mode — why do 96-byte runs write at 2.2 TB/s and 384-byte runs at 4.8?
TAG=${1:-pmc_ws}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
hipcc --offload-arch=gfx950 -O3 -o /tmp/write_streams.bin $REPO/tools/ubench/write_streams.hip || exit 1
cd /tmp
/tmp/write_streams.bin | tee $OUT/plain.txt
rocprofv3 -L 2>/dev/null | grep -oE "\bTCC_[A-Za-z0-9_]+" | sort -u > $OUT/tcc_counters.txt
pick() { for c in "$@"; do grep -qx "$c" $OUT/tcc_counters.txt && echo -n "$c "; done; }
SETS=("$(pick TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum)" "$(pick TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum)" "$(pick TCC_WRITEBACK_sum TCC_WRITE_sum TCC_READ_sum TCC_REQ_sum)" "$(pick TCC_EA0_WRREQ_STALL_sum TCC_EA_WRREQ_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_EA_WR_UNCACHED_32B_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum)" "$(pick TCC_HIT_sum TCC_MISS_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum)")
i=0
for SET in "${SETS[@]}"; do
  i=$((i+1)); [ -z "$SET" ] && continue
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o pmc -- /tmp/write_streams.bin > $OUT/p$i.log 2>&1
  echo "set $i ($SET) exit $?" | tee -a $OUT/summary.txt
done
python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob, collections
names = ["bucketed 96B", "bucketed 192B", "bucketed 384B", "bucketed 1.5KB", "aligned32 384B", "aligned32 768B", "dense", "granules"]
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    # dispatches in order: (flat 0/1) x 8 modes x 7 launches
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    pos = {d: k for k, d in enumerate(ids)}
    for r in rows:
        k = pos[int(r["Dispatch_Id"])]
        by[k // 7][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for g in sorted(by):
        print(("paired " if g < 8 else "flat   ") + "%-16s" % names[g % 8], {c: round(sum(v[2:]) / max(len(v[2:]), 1)) for c, v in by[g].items()})
PY
find $OUT -name "*.csv" -size +2M -delete
