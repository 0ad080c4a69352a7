#!/bin/bash
# round 4, call K: the whole GPU suite and the default bench with the prior's convolution / attention kernels; TunableOp experiment
mkdir -p gpurun_out/k
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/k/pytest.txt
T0=$SECONDS
timeout 900 python bench.py > gpurun_out/k/bench_default.json 2> gpurun_out/k/bench_default.err
echo "bench wall $((SECONDS-T0)) s" > gpurun_out/k/summary.txt
python - <<PY >> gpurun_out/k/summary.txt
import json
d=json.load(open("gpurun_out/k/bench_default.json"))
print({k:d.get(k) for k in ("value","ms_per_step","phases","iters_per_sec_nerf_only","iters_per_sec_without_unet")})
print("roofline", {k:d["roofline"].get(k) for k in ("achieved","frac","avg_launch_us","points_per_launch")})
PY
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/k/tunableop.csv UNET_AB_ONLY=11 timeout 400 python tools/unet_ab.py > gpurun_out/k/unet_tunableop.txt 2>&1
tail -4 gpurun_out/k/pytest.txt; cat gpurun_out/k/summary.txt; tail -3 gpurun_out/k/unet_tunableop.txt; tail -3 gpurun_out/k/bench_default.err
