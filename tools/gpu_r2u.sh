#!/bin/bash
TAG=${1:-r2u}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
for W in event query poll; do
  echo "-- SDFX_COUNT_WAIT=$W" | tee -a $OUT/summary.txt
  SDFX_COUNT_WAIT=$W timeout 600 python tools/step_timeline.py latent 2>$OUT/tl_$W.err | tail -7 | cut -c1-120 | tee -a $OUT/summary.txt
done
