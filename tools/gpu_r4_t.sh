#!/bin/bash
# round 4, call T: SQ counters of the attention and halo-convolution kernels (two passes), reduced to per-kernel averages
OUT=$PWD/gpurun_out/t; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
cd /tmp
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -o pmc -- python $REPO/tools/pmc_prior_kernels.py > $OUT/p1.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $OUT/p2 -o pmc -- python $REPO/tools/pmc_prior_kernels.py > $OUT/p2.log 2>&1
cd $REPO
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for key in ("k_attn_fwd<40", "k_attn_fwd<80", "k_attn_fwdILi40", "k_attn_fwdILi80", "k_conv3x3_halo", "k_conv_reduce"):
            if key in n:
                per[key + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(per.items()):
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())}, "launches", max(len(v) for v in cs.values()))
PY
rm -rf $OUT/p1 $OUT/p2
