#!/bin/bash
# scatter: run-folding threshold (SDFX_GRIDBWD_MERGE_RES: levels of resolution <= r fold lane runs before binning) on the current K1
TAG=${1:-scm}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp SDFX_DEV=1 SDFX_LIB=$PWD/stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so
timeout 900 python tools/scatter_ab.py 2 10 SDFX_GRIDBWD_MERGE_RES=640,450,330,230,170 2>&1 | grep -v amdgpu.ids | tee $OUT/merge_res.txt | grep -v round | cut -c1-300
