#!/bin/bash
# Ordered kernel list of ONE replayed iteration (latent phase, synthetic prior): what is still launched per step, how long each
# takes and the gaps between them.  Usage: bash tools/gpu_iter_trace.sh <tag> [phase]
TAG=${1:-it}; PHASE=${2:-latent}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --steps 12 --warmup 3 --phase $PHASE --guidance synthetic --no-cpu-baseline --no-kernel-bench --no-reference-flow > $OUT/prof.log 2>&1
echo "rocprof exit $?"
python3 - <<PY > $OUT/iteration_trace.txt
import csv, glob
f = glob.glob("$OUT/prof/**/bench_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
idx = [i for i, r in enumerate(rows) if "k_adan_update" in r["Kernel_Name"]]
print("kernels", len(rows), "optimiser steps", len(idx))
# a REPLAYED iteration of the timed region: bench.py ends with an eager kernel-timing pass whose launches are preceded by
# at::cuda::spin_kernel — take the last iteration before the first of those that holds neither a spin nor an occupancy refresh
spin = [i for i, r in enumerate(rows) if "spin_kernel" in r["Kernel_Name"]]
last = len(idx) - 2
if spin:
    last = max(k for k in range(len(idx) - 1) if idx[k + 1] < spin[0]) - 1
while last > 0 and any("k_occ_points" in r["Kernel_Name"] for r in rows[idx[last] + 1:idx[last + 1] + 1]):
    last -= 1
a, b = idx[last] + 1, idx[last + 1] + 1
t0 = int(rows[a]["Start_Timestamp"])
busy = 0
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print("%9.1f us  +%6.1f gap  %7.1f us  q%s  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name(r)[:110]))
    prev_end = max(prev_end, e)
print("launches %d  span %.1f us  busy %.1f us" % (b - a, (int(rows[b - 1]["End_Timestamp"]) - t0) / 1e3, busy / 1e3))
occ = [i for i, r in enumerate(rows) if "k_occ_points" in r["Kernel_Name"]]
if occ:
    import collections
    o = occ[-1]
    a2 = max(i for i in idx if i < o) + 1
    b2 = min(i for i in idx if i > o) + 1
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows[a2:b2]:
        k = agg[name(r)[:70]]; k[0] += 1; k[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("iteration with an occupancy refresh: launches %d  span %.1f us; top kernels:" % (b2 - a2, (int(rows[b2 - 1]["End_Timestamp"]) - int(rows[a2]["Start_Timestamp"])) / 1e3))
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   %8.1f us  x%-3d %s" % (t, n, k))
ends = [int(rows[i]["End_Timestamp"]) for i in idx]
hi = last + 1
periods = [(ends[k + 1] - ends[k]) / 1e3 for k in range(max(hi - 11, 0), hi)]
idle = [(int(rows[idx[k] + 1]["Start_Timestamp"]) - ends[k]) / 1e3 for k in range(max(hi - 11, 0), hi)]
print("iteration period (optimiser update to optimiser update), the 11 replayed iterations before it: median %.1f us; idle before the first kernel of an iteration: median %.1f us"
      % (sorted(periods)[len(periods) // 2], sorted(idle)[len(idle) // 2]))
PY
tail -8 $OUT/iteration_trace.txt
find $OUT/prof -type f -size +1M -delete 2>/dev/null
