#!/bin/bash
# last call of a round: the GPU test suite and smoke() on the final library
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== $(date) tag=$TAG" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest.txt 2>&1; grep -E "^E  |passed|failed|^FAILED" $OUT/pytest.txt | cut -c1-400 | head -20 | tee -a $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit: $?" | tee -a $OUT/summary.txt
tail -3 $OUT/smoke.log | cut -c1-300 | tee -a $OUT/summary.txt
