#!/bin/bash
# Why do the hashed levels 5-11 of the encode forward take 29-35 us each alone on the whole GPU? SQ / TCP / TCC counters of
# k_grid_fwd on ISOLATED levels (devtools library, SDFX_GRID_ONLY_LEVEL = l: the launch evaluates level l alone, spread over all
# eight XCDs) on the iteration's workload (7-point stencil batch of one 4096-ray view, B = 1.81 M points), one rocprofv3 --pmc pass
# per counter set and level (separate passes, kernel-trace only).   gpurun -- 'bash tools/gpu_pmc_fwd_levels.sh <tag> [levels]'
TAG=${1:-pmc_levels}; LEVELS=${2:-"0 2 4 5 7 9 11 13 15"}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
export SDFX_LIB=$REPO/stable-dreamfusion_amd/csrc/libsdfx_hip_dev.so
cd /tmp
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
      "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum"
      "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum")
for L in $LEVELS; do
  export SDFX_GRID_ONLY_LEVEL=$L
  # wall time of the isolated level without counters (event-timed by the tool)
  timeout 120 python $REPO/tools/encode_bench.py stencil f16 5 1,1,1 2>/dev/null | tail -1 | sed "s/^/level $L: /" | tee -a $OUT/times.txt
  i=0
  for SET in "${SETS[@]}"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/l${L}_p$i -o pmc -- python $REPO/tools/encode_bench.py stencil f16 2 1,1,1 > $OUT/l${L}_p$i.log 2>&1
    echo "level $L set $i exit $?" >> $OUT/log.txt
  done
done
unset SDFX_GRID_ONLY_LEVEL
find $OUT -type f -size +1M -delete 2>/dev/null
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections, re
rows = {}
for f in sorted(glob.glob("$OUT/l*_p*/*counter_collection.csv")):
    lvl = int(re.search(r"/l(\d+)_p", f).group(1))
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_grid_fwd" in r.get("Kernel_Name", ""):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        rows.setdefault(lvl, {})[k] = sum(v[-2:]) / max(len(v[-2:]), 1)    # the last launches (warm)
names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU",
         "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA",
         "TCP_PENDING_STALL_CYCLES_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TA_BUSY_avr", "TA_TA_BUSY_sum", "TCC_HIT_sum",
         "TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum"]
print("raw counters per launch (isolated level, B = 1.81 M stencil points):")
print("level " + " ".join(names))
for lvl in sorted(rows):
    print(lvl, " ".join("%.4g" % rows[lvl].get(n, float("nan")) for n in names))
print()
print("derived, per level: waves; cycles a wave is resident; of those waiting for anything / for an instruction dependency; VALU, VMEM-read,")
print("SALU instructions per wave; busy cycles of VALU / VMEM issue per wave; texture-address busy share of the kernel; distinct-line accesses per wave;")
print("L1 -> L2 read requests per wave; L2 hit rate; TCP cycles stalled on pending misses per wave")
for lvl in sorted(rows):
    r = rows[lvl]; g = lambda k: r.get(k, float("nan"))
    w = g("SQ_WAVES")
    print("L%-2d waves %.0f  cyc/wave %.0f  wait_any %.2f  wait_inst %.2f  active_inst %.2f | valu/wave %.0f vmem_rd/wave %.1f salu/wave %.0f | valu_busy/wave %.0f vmem_busy/wave %.0f | "
          "TA busy %.2f | tcp lines/wave %.1f  l2 reads/wave %.1f  l2 hit %.3f  tcp pending-stall/wave %.0f"
          % (lvl, w, g("SQ_WAVE_CYCLES") / w, g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
             g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_INSTS_VALU") / w, g("SQ_INSTS_VMEM_RD") / w, g("SQ_INSTS_SALU") / w,
             g("SQ_ACTIVE_INST_VALU") / w, g("SQ_ACTIVE_INST_VMEM") / w, g("TA_BUSY_avr") / (g("GRBM_GUI_ACTIVE") / 8.0) if g("GRBM_GUI_ACTIVE") == g("GRBM_GUI_ACTIVE") else float("nan"),
             g("TCP_TOTAL_CACHE_ACCESSES_sum") / w, g("TCP_TCC_READ_REQ_sum") / w, g("TCC_HIT_sum") / max(g("TCC_REQ_sum"), 1), g("TCP_PENDING_STALL_CYCLES_sum") / w))
PY
cat $OUT/times.txt
du -sh $OUT
