#!/bin/bash
# round 4, call N: the halo form of the convolution — tests, per-shape GPU times against the general kernel and MIOpen, UNet A/B
mkdir -p gpurun_out/n
timeout 300 python -m pytest tests/test_gpu_10_prior_kernels.py -m gpu -q -k "conv3x3" 2>&1 | tail -25 > gpurun_out/n/test.txt
timeout 400 python tools/conv_bench.py --sweep > gpurun_out/n/conv_bench.txt 2>&1
UNET_AB_ONLY=11 timeout 200 python tools/unet_ab.py 2>&1 | grep -v amdgpu > gpurun_out/n/unet.txt
tail -4 gpurun_out/n/test.txt; grep -v amdgpu gpurun_out/n/conv_bench.txt | cut -c1-220; cat gpurun_out/n/unet.txt
