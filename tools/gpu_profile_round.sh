#!/bin/bash
# Evidence for profiles/: (1) rocprofv3 kernel stats of the default bench, (2) separate PMC passes (FETCH_SIZE, WRITE_SIZE)
# over the same command for the dominant kernel, reduced to a small JSON that bench.py reports as roofline.traffic.
# Usage on the GPU box:  bash tools/gpu_profile_round.sh <tag>
TAG=${1:-round}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
CMD="python $REPO/bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-kernel-bench --no-reference-flow --no-nerf-only --no-children"
# The counter passes serialise every kernel: with the 860 M-parameter UNet of the default prior (thousands of stock
# PyTorch kernels per iteration, plus MIOpen's one-off solver search) one pass takes > 10 minutes. The kernels of this
# repository are the same with the synthetic prior, so the PMC passes run that.
PMC_CMD="$CMD --guidance synthetic"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $CMD > $OUT/stats.log 2>&1
echo "stats exit $?" | tee $OUT/summary.txt
tail -1 $OUT/stats.log | cut -c1-300 | tee -a $OUT/summary.txt
# the same pass with the synthetic prior: this repository's kernels only (what the PMC passes below run); kept under its own name
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_synth -o bench -- $PMC_CMD > $OUT/stats_synth.log 2>&1
echo "stats (synthetic prior) exit $?" | tee -a $OUT/summary.txt
cp $OUT/stats_synth/bench_kernel_stats.csv $OUT/kernel_stats_synthetic_prior.csv 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- $PMC_CMD > $OUT/pmc_$C.log 2>&1
  echo "pmc $C exit $?" | tee -a $OUT/summary.txt
done
# the gather roofline of the encode forward (round 4): 128-byte lines the vector-memory pipe looked up (TCP_TOTAL_CACHE_ACCESSES),
# lines that went on to L2 (TCP_TCC_READ_REQ), and how busy the texture-address units were (TA_BUSY over GRBM_GUI_ACTIVE)
i=0
for SET in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc_gather$i -o pmc -- $PMC_CMD > $OUT/pmc_gather$i.log 2>&1
  echo "pmc gather set $i ($SET) exit $?" | tee -a $OUT/summary.txt
done
python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob, json, collections
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/pmc_%s/*counter_collection.csv" % C):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != C:
                continue
            n = r["Kernel_Name"]
            for key in ("k_grid_fwd", "k_grid_forward", "k_grid_bwd_bin", "k_grid_bwd_reduce", "k_field_backward_nat", "k_field_forward_nat",
                        "k_render_train_fwd", "k_render_train_bwd", "k_composite_train_fwd", "k_composite_train_bwd", "k_march_count",
                        "k_adan_update", "k_grid_bwd_spill", "k_gn_stats", "k_gn_apply", "k_gn_bwd"):
                if key in n:
                    per[key].append(float(r["Counter_Value"]))
        for k, v in per.items():
            out.setdefault(k, {})[C + "_KB_avg"] = sum(v) / len(v)
            out[k]["launches"] = len(v)
for f in glob.glob("$OUT/pmc_gather*/*counter_collection.csv"):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        for key in ("k_grid_fwd", "k_grid_bwd_bin", "k_grid_bwd_reduce"):
            if key in r["Kernel_Name"]:
                per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in per.items():
        for c, v in cs.items():
            out.setdefault(k, {})[c + "_avg"] = sum(v) / len(v)
# the work of the launches the counters were averaged over, from the bytes each kernel WRITES per unit (bench.py scales the
# traffic per unit to the launch size it reports): encode forward 16 levels x 2 halves = 64 B per point; fused render forward one
# float weight per sample (+ 28 B per ray), backward 7 + 3 floats per sample
units = {"k_grid_fwd": (64.0, 0.0), "k_grid_forward": (64.0, 0.0), "k_render_train_fwd": (4.0, 4096 * 28.0), "k_render_train_bwd": (40.0, 0.0)}
for k, (per_unit, fixed) in units.items():
    if k in out and "WRITE_SIZE_KB_avg" in out[k]:
        out[k]["points_per_launch"] = max((out[k]["WRITE_SIZE_KB_avg"] * 1024.0 - fixed) / per_unit, 0.0)
        out[k]["points_per_launch_from"] = "WRITE_SIZE / %g B per unit" % per_unit
json.dump(out, open("$OUT/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
cp $OUT/stats/bench_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
python3 - <<PY | tee -a $OUT/summary.txt
import csv, collections
rows = list(csv.DictReader(open("$OUT/stats/bench_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "copyBuffer" in r["Kernel_Name"]]
d.sort()
print("copyBuffer: n=%d median=%.1f us p90=%.1f us max=%.1f us sum=%.1f ms; >100us: %d" % (len(d), d[len(d)//2], d[int(len(d)*0.9)], d[-1], sum(d)/1e3, sum(1 for x in d if x > 100)))
# what runs right before / after the long ones
long = [i for i, r in enumerate(rows) if "copyBuffer" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100000]
for i in long[-4:]:
    print("  long copy %.0f us, stream %s; prev: %s | next: %s" % ((int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3, rows[i].get("Stream_Id", rows[i].get("Queue_Id", "?")), rows[i-1]["Kernel_Name"][:40], rows[i+1]["Kernel_Name"][:40]))
PY
# bench.py's roofline figure against THIS trace: the bench re-runs its last stencil encode as 10 back-to-back launches of a replayed graph
# (6 replays: 60 launches in a row, nothing between them) — the same launches are in the kernel trace of the same process
python3 - <<PY | tee $OUT/encode_replay_in_trace.txt | tee -a $OUT/summary.txt
import csv, json
for tag, trace, log in (("default (SD-1.5-shaped prior)", "$OUT/stats/bench_kernel_trace.csv", "$OUT/stats.log"), ("synthetic prior", "$OUT/stats_synth/bench_kernel_trace.csv", "$OUT/stats_synth.log")):
    try:
        rows = list(csv.DictReader(open(trace)))
    except OSError:
        continue
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    best, cur = [], []
    for r in rows:
        if "k_grid_fwd" in r["Kernel_Name"]:
            cur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        else:
            if len(cur) > len(best): best = cur
            cur = []
    if len(cur) > len(best): best = cur
    line = None
    for l in open(log):
        if l.startswith("{"): line = l
    rf = json.loads(line)["roofline"] if line else {}
    d = sorted(best[-50:]) if best else [0.0]
    print("%s: longest streak of consecutive encode launches in the rocprofv3 kernel trace: n = %d, last 50: median %.1f us, min %.1f, max %.1f" % (tag, len(best), d[len(d) // 2], d[0], d[-1]))
    if rf:
        print("   the same process's bench line: roofline.avg_launch_us %.1f at %.0f points per launch -> frac %.4f; trace median -> %.1f ps per point = %.4f of 8 TB/s at 588 B per point"
              % (rf["avg_launch_us"], rf["points_per_launch"], rf["frac"], d[len(d) // 2] * 1e6 / rf["points_per_launch"], 588.0 * rf["points_per_launch"] / (d[len(d) // 2] * 1e-6) / 8e12))
PY
find $OUT -type f -size +1M -delete 2>/dev/null
du -sh $OUT
