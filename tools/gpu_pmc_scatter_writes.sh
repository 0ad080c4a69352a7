#!/bin/bash
# How K1 / K2 of the binned scatter talk to HBM: full-line (64-byte) against partial write requests, read requests by size.
# (K1 fetches 2 x 0.38 GB per launch at 3.27 M points against ~0.3 GB of inputs: is the rest read-modify-write of partially
# written list lines? runs of ~8 twelve-byte items never tile 128-byte lines.)
TAG=${1:-pmc_scatter_writes}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\bTCC_[A-Za-z0-9_]+" | sort -u > $OUT/tcc_counters.txt
grep -E "WRREQ|RDREQ|WRITEBACK|ATOMIC|EA0?_WR|EA0?_RD" $OUT/tcc_counters.txt | tr '\n' ' ' | tee $OUT/summary.txt; echo | tee -a $OUT/summary.txt
pick() { for c in "$@"; do grep -qx "$c" $OUT/tcc_counters.txt && echo -n "$c "; done; }
SETS=("$(pick TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum)" "$(pick TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum)" "$(pick TCC_WRITEBACK_sum TCC_WRITE_sum TCC_READ_sum TCC_REQ_sum)")
i=0
for SET in "${SETS[@]}"; do
  i=$((i+1)); [ -z "$SET" ] && continue
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o pmc -- python $REPO/tools/gridbwd_bench.py 5 > $OUT/p$i.log 2>&1
  echo "set $i ($SET) exit $?" | tee -a $OUT/summary.txt
done
python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        for key in ("k_grid_bwd_bin", "k_grid_bwd_reduce"):
            if key in r.get("Kernel_Name", ""):
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(k, {c: round(sum(v[-4:]) / max(len(v[-4:]), 1)) for c, v in cs.items()})
PY
find $OUT -type f -size +1M -delete 2>/dev/null
