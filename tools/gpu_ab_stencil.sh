#!/bin/bash
# A/B of the stencil source (sdfx_set_stencil_source) on the synthetic-prior iteration + kernel trace of both.
TAG=${1:-ab_stencil}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
python -m pytest tests/test_gpu_02_parity.py tests/test_gpu_05_trainer.py tests/test_gpu_03_render.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-500 | tee $OUT/tests.log
for V in 0 1; do
  SDFX_DEV=1 SDFX_STENCIL_SOURCE=$V python bench.py --steps 40 --warmup 8 --guidance synthetic --no-cpu-baseline --no-kernel-bench --no-reference-flow > $OUT/bench_src$V.json 2> $OUT/bench_src$V.err
  python tools/pick_bench.py < $OUT/bench_src$V.json 2>&1 | cut -c1-400 | tee -a $OUT/summary.txt
  ( cd /tmp && SDFX_DEV=1 SDFX_STENCIL_SOURCE=$V timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$V -o bench -- python $REPO/bench.py --steps 40 --warmup 8 --guidance synthetic --no-cpu-baseline --no-kernel-bench --no-reference-flow --no-nerf-only > $OUT/prof$V.log 2>&1 )
  f=$(find $OUT/prof$V -name "*kernel_stats.csv" | head -1)
  echo "== kernel stats SDFX_STENCIL_SOURCE=$V" | tee -a $OUT/summary.txt
  head -14 $f | cut -d, -f1-5 | cut -c1-150 | tee -a $OUT/summary.txt
  cp $f $OUT/kernel_stats_src$V.csv
  find $OUT/prof$V -type f -size +1M -delete
done
