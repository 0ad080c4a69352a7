#!/bin/bash
# Round 5's GPU-box calls, ONE parameterised script: gpurun -- 'bash tools/gpu_round5.sh <tag> <step> [<step> ...]'
# Steps (outputs under gpurun_out/<tag>/):
#   tests      pytest -m gpu (product library), smoke()
#   devtests   pytest -m gpu once more on the devtools library (the A/B tests that are skipped on the product library)
#   ubench     tools/ubench/write_streams.bin (list-write patterns of the scatter's first kernel)
#   timeline   tools/xcd_timeline.py: per-XCD busy timelines of the encode forward and the scatter, K1 ablations (devtools library)
#   bench      default bench.py line (N = 1)
#   stats      rocprofv3 --kernel-trace --stats of a short default bench -> kernel_stats csv (top rows)
#   pmc        tools/gpu_profile_round.sh-style FETCH_SIZE / WRITE_SIZE passes (synthetic prior) -> pmc_traffic.json
#   variants   A/B libraries under ab/ (tools/build_variant.py): scatter timeline + scatter tests on each
#   mergeres   scatter by SDFX_GRIDBWD_MERGE_RES; fwdvariants: encode forward by tiles per workgroup (devtools switches)
#   scatter    tools/gridbwd_bench.py 20 (standalone loop of K1 + K2 + K3) on the product library [and on ab/libsdfx_hip_base.so]
TAG=${1:-r5}; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
DEVLIB=$REPO/stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so
for STEP in "$@"; do
  echo "== step $STEP ($(date +%T))" | tee -a $OUT/summary.txt
  case $STEP in
    tests)
      timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -rs 2>&1 | tail -40 > $OUT/pytest_gpu.log
      tail -12 $OUT/pytest_gpu.log | cut -c1-300 | tee -a $OUT/summary.txt
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit: $?" | tee -a $OUT/summary.txt ;;
    devtests)
      SDFX_LIB=$DEVLIB timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -rs 2>&1 | tail -30 > $OUT/pytest_gpu_devlib.log
      tail -8 $OUT/pytest_gpu_devlib.log | cut -c1-300 | tee -a $OUT/summary.txt ;;
    ubench)
      timeout 300 tools/ubench/write_streams.bin ${TILES:-6400} 2>&1 | tee $OUT/write_streams.txt | tail -20 | tee -a $OUT/summary.txt ;;
    timeline)
      SDFX_LIB=$DEVLIB timeout 600 python tools/xcd_timeline.py ${VIEWS:-2} ${ABLATE:-1} 2>&1 | grep -v "amdgpu.ids" | tee $OUT/xcd_timeline.txt | cut -c1-220 | tee -a $OUT/summary.txt ;;
    bench)
      timeout 900 python bench.py ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit: $?" | tee -a $OUT/summary.txt
      tail -4 $OUT/bench.err | tee -a $OUT/summary.txt
      python tools/pick_bench.py < $OUT/bench.json 2>&1 | tee -a $OUT/summary.txt ;;
    stats)
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-kernel-bench --no-nerf-only --no-reference-flow --no-children ${STATS_ARGS:-} > $OUT/prof.log 2>&1 ); echo "rocprof exit: $?" | tee -a $OUT/summary.txt
      f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 $f > $OUT/kernel_stats_top40.csv && head -14 $f | cut -c1-200 | tee -a $OUT/summary.txt
      find $OUT/prof -type f -size +2M -delete 2>/dev/null ;;
    pmc)
      bash tools/gpu_profile_round.sh $TAG/pmc 2>&1 | tail -30 | tee -a $OUT/summary.txt ;;
    scatter)
      timeout 300 python tools/gridbwd_bench.py 20 2>&1 | grep -v "amdgpu.ids" | tee $OUT/gridbwd_bench.txt | tee -a $OUT/summary.txt
      if [ -f ab/libsdfx_hip_base.so ]; then SDFX_LIB=$REPO/ab/libsdfx_hip_base.so timeout 300 python tools/gridbwd_bench.py 20 2>&1 | grep -v "amdgpu.ids" | sed 's/^/BASE: /' | tee -a $OUT/gridbwd_bench.txt | tee -a $OUT/summary.txt; fi ;;
    variants)   # every ab/libsdfx_hip_<name>.so built by tools/build_variant.py (VARIANTS="name ..." picks some): timeline without ablations + scatter tests
      for lib in ${VARIANTS:-$(ls ab/libsdfx_hip_*.so | sed 's#ab/libsdfx_hip_##; s#\.so##' | grep -v '^base$')}; do
        echo "== variant $lib" | tee -a $OUT/summary.txt
        SDFX_LIB=$REPO/ab/libsdfx_hip_$lib.so timeout 300 python tools/xcd_timeline.py ${VIEWS:-2} 0 2>&1 | grep -v "amdgpu.ids" > $OUT/xcd_timeline_$lib.txt
        grep -E "^encode|^scatter|^== scatter|XCD finish" $OUT/xcd_timeline_$lib.txt | cut -c1-200 | tee -a $OUT/summary.txt
        SDFX_LIB=$REPO/ab/libsdfx_hip_$lib.so timeout 600 python -m pytest tests/test_gpu_02_parity.py tests/test_gpu_zz_stress.py tests/test_gpu_00_vs_reference_kernels.py -m gpu -q -p no:cacheprovider -k "grid or scatter or stencil or binned" 2>&1 | tail -3 | cut -c1-300 | tee -a $OUT/summary.txt
      done ;;
    mergeres)   # K1's run folding by level resolution (devtools SDFX_GRIDBWD_MERGE_RES; product: 640): scatter time, K1 / K2 spans
      for r in ${MERGE_RES:-0 213 409 640 1023}; do
        echo "== SDFX_GRIDBWD_MERGE_RES=$r" | tee -a $OUT/summary.txt
        SDFX_LIB=$DEVLIB SDFX_GRIDBWD_MERGE_RES=$r FWD_LEVELS=0 timeout 300 python tools/xcd_timeline.py ${VIEWS:-2} 0 2>&1 | grep -E "^scatter \\(|^== scatter" | cut -c1-160 | tee -a $OUT/summary.txt
      done ;;
    fwdscalar)
      SDFX_LIB=$DEVLIB FWD_SCALAR=${FWD_SCALAR:-0,100000} FWD_LEVELS=0 timeout 300 python tools/xcd_timeline.py ${VIEWS:-2} 0 2>&1 | grep -E "scalar below|every level alone" | tee -a $OUT/summary.txt ;;
    fwdscalarfrom)
      SDFX_LIB=$DEVLIB FWD_LEVELS=0 timeout 300 python tools/xcd_timeline.py ${VIEWS:-2} 0 2>&1 | grep -v "amdgpu.ids" > $OUT/xcd_timeline_scalar_from.txt
      grep -E "round [0-9] from|XCD finish|4-byte gathers" $OUT/xcd_timeline_scalar_from.txt | cut -c1-220 | tee -a $OUT/summary.txt ;;
    fwdvariants)
      SDFX_LIB=$DEVLIB FWD_VARIANTS=1 FWD_LEVELS=0 timeout 300 python tools/xcd_timeline.py ${VIEWS:-2} 0 2>&1 | grep -E "tiles per workgroup" | tee -a $OUT/summary.txt ;;
    *) echo "unknown step $STEP" | tee -a $OUT/summary.txt ;;
  esac
done
du -sh $OUT | tee -a $OUT/summary.txt
