#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c60
mkdir -p $OUT
cd $ROOT
MCBRAT_FLIGHT_FUZZ=${1:-12} timeout -k 10 1000 python -m pytest tests/test_gpu_block_walk.py -v -m gpu --timeout 90 --timeout-method thread > $OUT/tests.log 2>&1; rc=$?
grep -c PASSED $OUT/tests.log; grep -n "FAILED\|Timeout\|^E   *Assert" $OUT/tests.log | head -20; tail -3 $OUT/tests.log
echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python scripts/quick_bench.py --case step --ppb 100000 --batches 100 --thr 16 --reps 3 > $OUT/step.log 2>&1; grep case= $OUT/step.log
