#!/bin/bash
# Does the default timed region apply every optimiser step? Reduced bench runs on the product library and, if present, on a saved
# previous build (ab/libsdfx_hip_base.so): value, optimizer_steps_applied, grad_scale per run.  gpurun -- 'bash tools/gpu_applied_check.sh <tag> [runs]'
TAG=${1:-applied}; RUNS=${2:-2}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
ARGS="--no-stock-prior --no-children --no-cpu-baseline --no-kernel-bench --no-nerf-only --no-reference-flow"
pick() { python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2], 'latent %.1f' % d['phases']['latent']['iters_per_sec'], 'value %.1f applied %s scale %s samples %.0f frac %.3f enc_us %.1f pts %.0f' % (d['value'], d['optimizer_steps_applied'], d['grad_scale'], d['samples_per_iter'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['points_per_launch']))" $1 $2; }
for i in $(seq 1 $RUNS); do
  timeout 300 python bench.py $ARGS > $OUT/product_$i.json 2> $OUT/product_$i.err; pick $OUT/product_$i.json product_$i | tee -a $OUT/summary.txt
done
if [ -f ab/libsdfx_hip_base.so ]; then
  for i in $(seq 1 $RUNS); do
    SDFX_LIB=$PWD/ab/libsdfx_hip_base.so timeout 300 python bench.py $ARGS > $OUT/base_$i.json 2> $OUT/base_$i.err; pick $OUT/base_$i.json base_$i | tee -a $OUT/summary.txt
  done
fi
