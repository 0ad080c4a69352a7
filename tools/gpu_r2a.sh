#!/bin/bash
# Round-2 first GPU call: un-gated parity tests, the RGB phase and the reference host flow of bench.py,
# and the wave-per-ray counting pass timed against the thread-per-ray one.
TAG=${1:-r2a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
SDFX_TEST_EXPERIMENTAL=1 timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -150 > $OUT/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" | tee -a $OUT/summary.txt
tail -40 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
# RGB phase (after the latent warm-up): VAE encoder with gradient, lambertian / textureless, random backgrounds
timeout 600 python bench.py --steps 20 --warmup 5 --phase rgb --no-cpu-baseline --no-kernel-bench > $OUT/bench_rgb.json 2> $OUT/bench_rgb.err
echo "bench rgb exit: $?" | tee -a $OUT/summary.txt
tail -5 $OUT/bench_rgb.err | tee -a $OUT/summary.txt
cat $OUT/bench_rgb.json | tee -a $OUT/summary.txt
# the reference's host flow (GradScaler + foreach Adan) on the same kernels, NeRF part only and with the UNet
SDFX_TRAIN_MODE=reference timeout 300 python bench.py --steps 20 --warmup 5 --guidance synthetic --no-cpu-baseline --no-kernel-bench > $OUT/bench_refflow_synth.json 2> $OUT/bench_refflow_synth.err
echo "bench refflow synth exit: $?" | tee -a $OUT/summary.txt
tail -3 $OUT/bench_refflow_synth.err | tee -a $OUT/summary.txt
cat $OUT/bench_refflow_synth.json | tee -a $OUT/summary.txt
# counting pass: thread-per-ray vs wave-per-ray
timeout 300 python tools/march_bench.py > $OUT/march_bench.txt 2>&1
echo "march bench exit: $?" | tee -a $OUT/summary.txt
cat $OUT/march_bench.txt | tee -a $OUT/summary.txt
