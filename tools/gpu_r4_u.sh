#!/bin/bash
# round 4, call U: the single-halo-buffer variant of the halo convolution (devtools library) — bit identity, per-shape times off / on / on with 768 workgroups
mkdir -p gpurun_out/u
export SDFX_LIB=$PWD/stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so
timeout 100 python -m pytest tests/test_gpu_10_prior_kernels.py -m gpu -q -k "single_buffer" 2>&1 | tail -4 > gpurun_out/u/test.txt
timeout 120 python tools/conv_bench.py 2>&1 | grep -v amdgpu | cut -c1-100 > gpurun_out/u/off.txt
SDFX_CONV_HALO_SINGLE=1 timeout 120 python tools/conv_bench.py 2>&1 | grep -v amdgpu | cut -c1-100 > gpurun_out/u/single.txt
SDFX_CONV_HALO_SINGLE=1 SDFX_CONV_HALO_TARGET=768 timeout 120 python tools/conv_bench.py 2>&1 | grep -v amdgpu | cut -c1-100 > gpurun_out/u/single768.txt
cat gpurun_out/u/test.txt; paste <(cut -c1-32,62-78 gpurun_out/u/off.txt) <(cut -c62-78 gpurun_out/u/single.txt) <(cut -c62-78 gpurun_out/u/single768.txt)
