#!/bin/bash
# round 4, call R: UNet evaluation, previous build vs this one, order base / new / base / new (box noise)
mkdir -p gpurun_out/r
for i in 1 2; do
SDFX_LIB=$PWD/ab/libsdfx_hip_base.so UNET_AB_ONLY=11 timeout 200 python tools/unet_ab.py 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/base $i: /" >> gpurun_out/r/unet.txt
UNET_AB_ONLY=11 timeout 200 python tools/unet_ab.py 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/new  $i: /" >> gpurun_out/r/unet.txt
done
timeout 200 python tools/gn_bench.py 2>&1 | grep -v amdgpu | tail -1 >> gpurun_out/r/unet.txt
cat gpurun_out/r/unet.txt
