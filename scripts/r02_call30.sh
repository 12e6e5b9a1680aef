#!/bin/bash
# clear-air flight: correctness subset, then A/B of layerSkip = 2 (layers only) against 1 (with the flight)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c30
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_layer_skip.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?
tail -15 $OUT/tests.log
if [ $rc -ne 0 ]; then echo "tests rc=$rc"; exit $rc; fi
for c in landsat radar; do
  timeout -k 10 300 python scripts/quick_bench.py --case $c --ppb 1000000 --batches 100 --thr 32 --skip 2 1 --counters >> $OUT/ab.log 2>&1 || exit 1
done
grep -E "case=|per photon|walk iters" $OUT/ab.log
echo finished
