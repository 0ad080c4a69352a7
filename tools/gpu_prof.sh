#!/bin/bash
# rocprofv3 kernel-stats pass of a short bench run; keeps only the small CSV summaries.
TAG=${1:-prof}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --steps ${STEPS:-8} --warmup 3 --no-cpu-baseline --no-kernel-bench > $OUT/prof.log 2>&1
echo "rocprof exit: $?" | tee $OUT/summary.txt
find $OUT/prof -type f -size +2M -delete 2>/dev/null
find $OUT/prof -type f | head -20 | tee -a $OUT/summary.txt
STATS=$(find $OUT/prof -name "*kernel_stats*" | head -1)
[ -n "$STATS" ] && head -40 "$STATS" | tee -a $OUT/summary.txt
tail -2 $OUT/prof.log | cut -c1-600 | tee -a $OUT/summary.txt
du -sh $OUT | tee -a $OUT/summary.txt
