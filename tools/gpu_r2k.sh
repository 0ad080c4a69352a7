#!/bin/bash
TAG=${1:-r2k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_infer.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -15 | cut -c1-260 | tee -a $OUT/summary.txt
for W in 2 3 4; do
  echo "SDFX_INFER_WAVES=$W" | tee -a $OUT/summary.txt
  SDFX_INFER_WAVES=$W HW=64,256,800 SCENES=blobs timeout 600 python tools/infer_bench.py 2>&1 | grep -v amdgpu.ids | grep persistent | tee -a $OUT/summary.txt
done
SDFX_INFER_WAVES=3 HW=800 SCENES=init timeout 600 python tools/infer_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
