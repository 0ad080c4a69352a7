#!/bin/bash
# encode forward, two levels per wave: parity tests of the product library, then the A/B + timelines on the devtools library
TAG=${1:-pair}; VIEWS=${2:-2}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_02_parity.py tests/test_gpu_00_vs_reference_kernels.py tests/test_gpu_06_occupancy.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-600 | tee $OUT/tests.log
export SDFX_DEV=1 SDFX_LIB=$PWD/stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so
timeout 900 python tools/pair_ab.py $VIEWS 10 2>&1 | tee $OUT/pair_ab.txt | grep -v '^   XCD [0-9]' | cut -c1-260
