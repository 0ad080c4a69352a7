#!/bin/bash
TAG=${1:-r2q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_trainer.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "rgb_phase" 2>&1 | tail -5 | cut -c1-250 | tee -a $OUT/summary.txt
bash tools/gpu_iter_trace.sh $TAG/lat latent 2>&1 | tail -4 | tee -a $OUT/summary.txt
bash tools/gpu_iter_trace.sh $TAG/rgb rgb 2>&1 | tail -4 | tee -a $OUT/summary.txt
