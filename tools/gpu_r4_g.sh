#!/bin/bash
# round 4, call G: render kernels against the sample count, the scatter's standalone loop under the kernel trace, the VAE
# wide-head attention A/B, and the default bench line with the gather roofline (profiles/r04_pmc_traffic.json present).
TAG=${1:-r4g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 300 python -m pytest tests/test_gpu_04_sds.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -1 | tee -a $OUT/summary.txt
bash tools/gpu_render_scaling.sh $TAG/render > /dev/null 2>&1; cat $OUT/render/summary.txt | tee -a $OUT/summary.txt
for i in 1 2; do timeout 200 python tools/gridbwd_bench.py 20 2>&1 | tail -2 | tee -a $OUT/summary.txt; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/scatter -o sc -- python $REPO/tools/gridbwd_bench.py 20 > $REPO/$OUT/scatter.log 2>&1 )
python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob
f = glob.glob("$OUT/scatter/**/sc_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("   %-60s calls %s avg %.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
Q="--steps 12 --warmup 4 --no-cpu-baseline --no-kernel-bench --no-nerf-only --no-reference-flow --phase rgb"
for V in "SDFX_WIDE_HEAD_MATMUL=0" "SDFX_WIDE_HEAD_MATMUL=1"; do
  env $V timeout 900 python bench.py $Q > $OUT/bench_$V.json 2> $OUT/bench_$V.err
  echo "rgb $V exit $?: $(python -c "import json,sys; d=json.load(open('$OUT/bench_$V.json')); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)" | tee -a $OUT/summary.txt
done
T0=$SECONDS
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench default exit: $? wall $((SECONDS-T0)) s" | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
d=json.load(open("$OUT/bench_default.json"))
print({k:d.get(k) for k in ("value","ms_per_step","phases","iters_per_sec_nerf_only","ms_nerf_only","xcd_round_robin")})
print("roofline", {k:d["roofline"].get(k) for k in ("achieved","frac","avg_launch_us","points_per_launch","hbm_frac")})
print("gather", d["roofline"].get("gather"))
PY
find $OUT -type f -size +1M -delete 2>/dev/null
du -sh $OUT | tee -a $OUT/summary.txt
