#!/bin/bash
# round 4, call J: attention after the register / packed-math fixes, fused QKV projections, kernel trace of one UNet evaluation
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_10_prior_kernels.py tests/test_gpu_04_sds.py -m gpu -q 2>&1 | tail -30 > gpurun_out/prior_kernels_test.txt
timeout 200 python tools/attn_bench.py > gpurun_out/attn_bench.txt 2>&1
timeout 300 python tools/unet_ab.py > gpurun_out/unet_ab.txt 2>&1
SDFX_QKV_FUSION=0 UNET_AB_ONLY=11 timeout 300 python tools/unet_ab.py > gpurun_out/unet_ab_noqkv.txt 2>&1
cd /tmp && export TMPDIR=/tmp
UNET_AB_ONLY=11 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/unet_prof -- python $GRAFT_REPO_ROOT/tools/unet_ab.py > $GRAFT_REPO_ROOT/gpurun_out/unet_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/unet_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/unet_kernel_stats.csv
find gpurun_out/unet_prof -type f ! -name "*kernel_stats.csv" -delete
tail -5 gpurun_out/prior_kernels_test.txt; cat gpurun_out/attn_bench.txt gpurun_out/unet_ab.txt gpurun_out/unet_ab_noqkv.txt | grep -v amdgpu.ids; head -40 gpurun_out/unet_kernel_stats.csv | cut -c1-160
