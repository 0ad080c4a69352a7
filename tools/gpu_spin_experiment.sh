#!/bin/bash
# does a GPU-side spin in front of the timed region's eager launches (round 5's KernelTimer) change how many
# optimiser steps the reduced default bench applies? Alternates runs with and without it on ONE box, with the per-step trace
# (global_step, samples, loss scale, applied, gradient norm). The pool has two box classes (latent phase ~106 it/s and ~92 it/s); round 5
# saw steps lost only on the slower one: the first run tells the class, and on a fast box only `PAIRS_FAST` pairs are run.
#   gpurun -- 'bash tools/gpu_spin_experiment.sh <tag> [pairs on a slow box] [pairs on a fast box]'
TAG=${1:-spin}; PAIRS_SLOW=${2:-5}; PAIRS_FAST=${3:-1}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
ARGS="--no-stock-prior --no-children --no-cpu-baseline --no-kernel-bench --no-nerf-only --no-reference-flow"
pick() { python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2], 'latent %.1f' % d['phases']['latent']['iters_per_sec'], 'value %.1f applied %s scale %s samples %.0f timed_region %s' % (d['value'], d['optimizer_steps_applied'], d['grad_scale'], d['samples_per_iter'], d['graph_stats_timed_region']))" $1 $2; }
one() {  # spin index
  SDFX_BENCH_TRACE=1 SDFX_BENCH_SPIN_TIMED=$1 timeout 300 python bench.py $ARGS > $OUT/spin$1_$2.json 2> $OUT/spin$1_$2.err
  pick $OUT/spin$1_$2.json spin$1_$2 | tee -a $OUT/summary.txt
  grep "^\[trace\]" $OUT/spin$1_$2.err | awk '$11 != prev && NR > 1 {print "   skipped at:", $0} {prev = $11}' | tee -a $OUT/summary.txt
}
one 1 1
LAT=$(python -c "
import json,sys
d=json.loads([l for l in open('$OUT/spin1_1.json') if l.startswith('{')][-1]); print(int(d['phases']['latent']['iters_per_sec']))")
if [ "$LAT" -lt 98 ]; then PAIRS=$PAIRS_SLOW; echo "slow box (latent $LAT it/s): $PAIRS pairs" | tee -a $OUT/summary.txt
else PAIRS=$PAIRS_FAST; echo "fast box (latent $LAT it/s): $PAIRS pairs" | tee -a $OUT/summary.txt; fi
one 0 1
for i in $(seq 2 $PAIRS); do one 1 $i; one 0 $i; done
