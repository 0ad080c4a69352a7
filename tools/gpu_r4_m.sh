#!/bin/bash
# round 4, call M: the one-tap (GEMM) form of the convolution kernel — tests and per-shape comparison with F.linear in graph replay
mkdir -p gpurun_out/m
timeout 300 python -m pytest tests/test_gpu_10_prior_kernels.py -m gpu -q -k linear 2>&1 | tail -12 > gpurun_out/m/test.txt
timeout 300 python tools/linear_bench.py > gpurun_out/m/linear_bench.txt 2>&1
tail -3 gpurun_out/m/test.txt; grep -v amdgpu gpurun_out/m/linear_bench.txt
