#!/bin/bash
TAG=${1:-r2m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 120 tools/ubench/launch_floor.bin 2>&1 | tee $OUT/launch_floor.txt | tee -a $OUT/summary.txt
for PH in latent rgb; do
  timeout 600 python bench.py --steps 40 --warmup 8 --phase $PH --guidance synthetic --no-cpu-baseline --no-kernel-bench > $OUT/bench_synth_$PH.json 2> $OUT/bench_synth_$PH.err
  python tools/pick_bench.py < $OUT/bench_synth_$PH.json 2>&1 | tee -a $OUT/summary.txt
done
REPO=$PWD; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o bench -- python $REPO/bench.py --steps 20 --warmup 5 --phase latent --guidance synthetic --no-cpu-baseline --no-kernel-bench --no-reference-flow > $REPO/$OUT/prof.log 2>&1 )
cp $OUT/prof/*/bench_kernel_stats.csv $OUT/kernel_stats_synth_latent.csv 2>/dev/null || find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_synth_latent.csv \;
find $OUT/prof -type f -size +1M -delete 2>/dev/null
head -30 $OUT/kernel_stats_synth_latent.csv | cut -c1-60,200-400 | head -5
