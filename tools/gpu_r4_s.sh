#!/bin/bash
# round 4, call S: the pipelined attention variant (devtools library) — bit identity, then the per-shape bench with it off / on
mkdir -p gpurun_out/s
export SDFX_LIB=$PWD/stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so
timeout 200 python -m pytest tests/test_gpu_10_prior_kernels.py -m gpu -q -k attention 2>&1 | tail -6 > gpurun_out/s/test.txt
timeout 100 python tools/attn_bench.py 2>&1 | grep -v amdgpu | head -4 > gpurun_out/s/bench_off.txt
SDFX_ATTN_PIPE=1 timeout 100 python tools/attn_bench.py 2>&1 | grep -v amdgpu | head -4 > gpurun_out/s/bench_on.txt
SDFX_ATTN_PIPE=2 timeout 100 python tools/attn_bench.py 2>&1 | grep -v amdgpu | head -4 > gpurun_out/s/bench_swz.txt
cat gpurun_out/s/test.txt gpurun_out/s/bench_off.txt gpurun_out/s/bench_on.txt gpurun_out/s/bench_swz.txt
