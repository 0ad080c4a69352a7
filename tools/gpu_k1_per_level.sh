#!/bin/bash
# True per-level cost of K1 / K2: the scatter restricted to max_level = 1..16 with the FLAT mapping (every level uses the whole GPU,
# so the increments add up), kernel-trace durations of each kernel.
TAG=${1:-k1lvl}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for F in 1 0; do
SDFX_GRIDBWD_FLAT=$F PER_LEVEL=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t$F -o t -- python $REPO/tools/gridbwd_bench.py 3 > $OUT/t$F.log 2>&1
python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob
f = glob.glob("$OUT/t$F/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for kn in ("k_grid_bwd_bin", "k_grid_bwd_reduce"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if kn in r["Kernel_Name"]]
    per = d[4:]                       # 4 full calls first (n + 1), then 4 calls per max_level
    last = [per[4 * i + 3] for i in range(len(per) // 4)]
    inc = [last[0]] + [last[i] - last[i - 1] for i in range(1, len(last))]
    print("FLAT=$F", kn, "full %.1f" % d[3], "cumulative", [round(x) for x in last])
    print("   increments", [round(x) for x in inc])
PY
done
find $OUT -type f -size +1M -delete 2>/dev/null
