#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c43
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_flight.py tests/test_gpu_layer_skip.py -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $OUT/tests.log
[ $rc -eq 0 ] || exit $rc
for c in landsat radar; do
for thr in 16 20 24; do
timeout -k 10 200 python scripts/ab_compare.py $c $thr >> $OUT/ab.log 2>&1 || exit 1
done
done
grep lib= $OUT/ab.log
echo finished
