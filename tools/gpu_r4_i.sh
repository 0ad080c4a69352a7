#!/bin/bash
# round 4, call I: attention + convolution kernels of the prior — tests, per-shape benches, ablation of the convolution's K step, UNet A/B
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_10_prior_kernels.py -m gpu -q 2>&1 | tail -30 > gpurun_out/prior_kernels_test.txt
timeout 200 python tools/attn_bench.py > gpurun_out/attn_bench.txt 2>&1
SDFX_LIB=$PWD/stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so timeout 200 python tools/conv_ablate.py > gpurun_out/conv_ablate.txt 2>&1
timeout 300 python tools/unet_ab.py > gpurun_out/unet_ab.txt 2>&1
tail -5 gpurun_out/prior_kernels_test.txt; cat gpurun_out/attn_bench.txt gpurun_out/conv_ablate.txt gpurun_out/unet_ab.txt | grep -v amdgpu.ids
