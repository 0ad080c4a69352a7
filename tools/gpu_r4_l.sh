#!/bin/bash
# round 4, call L: where TunableOp's 1 ms of the UNet evaluation comes from — rocBLAS vs hipBLASLt vs TunableOp with a ready file
mkdir -p gpurun_out/l
for B in cublas cublaslt; do UNET_BLAS=$B UNET_AB_ONLY=11 timeout 200 python tools/unet_ab.py 2>&1 | grep -v amdgpu > gpurun_out/l/blas_$B.txt; done
cp profiles/r04_tunableop_unet.csv /tmp/tun0.csv 2>/dev/null
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME=/tmp/tun.csv UNET_AB_ONLY=11 timeout 200 python tools/unet_ab.py 2>&1 | grep -v amdgpu > gpurun_out/l/tunable_notuning.txt
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/l/tunable_again.csv UNET_AB_ONLY=11 timeout 300 python tools/unet_ab.py 2>&1 | grep -v amdgpu > gpurun_out/l/tunable_tuning.txt
cd /tmp && export TMPDIR=/tmp
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME=/tmp/tun.csv UNET_AB_ONLY=11 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/l/prof -- python $GRAFT_REPO_ROOT/tools/unet_ab.py > $GRAFT_REPO_ROOT/gpurun_out/l/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/l/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/l/unet_kernel_stats_tunable.csv
rm -rf gpurun_out/l/prof
tail -2 gpurun_out/l/*.txt
