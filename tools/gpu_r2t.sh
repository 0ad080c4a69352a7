#!/bin/bash
TAG=${1:-r2t}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_sds.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -4 | cut -c1-250 | tee -a $OUT/summary.txt
timeout 600 python tools/step_timeline.py latent 2>$OUT/tl_lat.err | tail -8 | tee -a $OUT/summary.txt
timeout 600 python tools/step_timeline.py rgb 2>$OUT/tl_rgb.err | tail -8 | tee -a $OUT/summary.txt
tail -3 $OUT/tl_lat.err
