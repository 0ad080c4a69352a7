#!/bin/bash
# round 4, call O: GroupNorm kernels with batched loads — tests, per-shape GPU times against the previous build (ab/libsdfx_hip_base.so), UNet
mkdir -p gpurun_out/o
timeout 300 python -m pytest tests/test_gpu_04_sds.py tests/test_gpu_10_prior_kernels.py -m gpu -q 2>&1 | tail -8 > gpurun_out/o/test.txt
timeout 200 python tools/gn_bench.py 2>&1 | grep -v amdgpu > gpurun_out/o/gn_new.txt
SDFX_LIB=$PWD/ab/libsdfx_hip_base.so timeout 200 python tools/gn_bench.py 2>&1 | grep -v amdgpu > gpurun_out/o/gn_base.txt
UNET_AB_ONLY=11 timeout 200 python tools/unet_ab.py 2>&1 | grep -v amdgpu > gpurun_out/o/unet_new.txt
SDFX_LIB=$PWD/ab/libsdfx_hip_base.so UNET_AB_ONLY=11 timeout 200 python tools/unet_ab.py 2>&1 | grep -v amdgpu > gpurun_out/o/unet_base.txt
tail -3 gpurun_out/o/test.txt; paste gpurun_out/o/gn_base.txt gpurun_out/o/gn_new.txt | cut -c1-200; cat gpurun_out/o/unet_base.txt gpurun_out/o/unet_new.txt
