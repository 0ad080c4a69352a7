#!/bin/bash
TAG=${1:-r2v}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
run() {
  echo "-- $*" | tee -a $OUT/summary.txt
  env "$@" timeout 300 python tools/step_timeline.py latent 2>$OUT/err.txt | grep -E "period|train_done|count_done" | cut -c1-100 | tee -a $OUT/summary.txt
}
run A=0
run GPU_FORCE_QUEUE_PROFILING=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run SDFX_PREFETCH=0
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=4096
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
