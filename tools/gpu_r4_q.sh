#!/bin/bash
# round 4, call Q: GroupNorm one-launch up to 32 x 32 maps, halo convolution at 8 x 8 — tests, per-shape GPU times against the previous build, UNet
mkdir -p gpurun_out/q
timeout 300 python -m pytest tests/test_gpu_04_sds.py tests/test_gpu_10_prior_kernels.py -m gpu -q 2>&1 | tail -8 > gpurun_out/q/test.txt
timeout 200 python tools/gn_bench.py 2>&1 | grep -v amdgpu > gpurun_out/q/gn_new.txt
SDFX_LIB=$PWD/ab/libsdfx_hip_base.so timeout 200 python tools/gn_bench.py 2>&1 | grep -v amdgpu > gpurun_out/q/gn_base.txt
UNET_AB_ONLY=11 timeout 200 python tools/unet_ab.py 2>&1 | grep -v amdgpu > gpurun_out/q/unet_new.txt
SDFX_LIB=$PWD/ab/libsdfx_hip_base.so UNET_AB_ONLY=11 timeout 200 python tools/unet_ab.py 2>&1 | grep -v amdgpu > gpurun_out/q/unet_base.txt
tail -3 gpurun_out/q/test.txt; paste gpurun_out/q/gn_base.txt gpurun_out/q/gn_new.txt | cut -c1-200; cat gpurun_out/q/unet_base.txt gpurun_out/q/unet_new.txt
timeout 300 python tools/conv_bench.py > gpurun_out/q/conv_bench.txt 2>&1; grep -v amdgpu gpurun_out/q/conv_bench.txt | cut -c1-120
