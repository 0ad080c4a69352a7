#!/bin/bash
# pair plan with its measured price table: A/B at two batch sizes, the refresh's curve-ordered batch, then the GPU suite and a reduced bench
TAG=${1:-pair3}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
DEVLIB=$PWD/stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so
for V in 2 1; do
  SDFX_DEV=1 SDFX_LIB=$DEVLIB PAIR_TPW="2,4" timeout 900 python tools/pair_ab.py $V 10 2>&1 | tee $OUT/pair_ab_views$V.txt | grep -E 'min |spread|samples' | cut -c1-200
done
for P in 0 1; do
  echo "== refresh batch (morton), SDFX_GRID_PAIR=$P" | tee -a $OUT/refresh.txt
  SDFX_DEV=1 SDFX_LIB=$DEVLIB SDFX_GRID_PAIR=$P timeout 300 python tools/encode_bench.py morton f16 10 1,1,1 1,1,0 2>&1 | tail -2 | cut -c1-300 | tee -a $OUT/refresh.txt
  SDFX_DEV=1 SDFX_LIB=$DEVLIB SDFX_GRID_PAIR=$P timeout 300 python tools/encode_bench.py uniform f16 10 1,1,0 2>&1 | tail -1 | cut -c1-300 | tee -a $OUT/refresh.txt
done
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest.txt 2>&1; grep -E "^E  |passed|failed|^FAILED" $OUT/pytest.txt | cut -c1-400 | head -20 | tee $OUT/tests_summary.txt
python bench.py --steps 40 --warmup 8 --guidance synthetic --no-cpu-baseline --no-reference-flow > $OUT/bench_synth.json 2> $OUT/bench_synth.err
python tools/pick_bench.py < $OUT/bench_synth.json 2>&1 | cut -c1-400 | tee $OUT/bench_synth_summary.txt
