#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c42
mkdir -p $OUT
cd $ROOT
for c in landsat radar; do
for nu in 0 1 0 1; do
MCBRAT_NEAR_UNIFORM_WALK=$nu timeout -k 10 200 python scripts/ab_compare.py $c 20 >> $OUT/ab.log 2>&1 || exit 1
echo "   (near-uniform walk = $nu)" >> $OUT/ab.log
done
done
grep -A1 lib= $OUT/ab.log
MCBRAT_NEAR_UNIFORM_WALK=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_flight.py tests/test_gpu_layer_skip.py tests/test_gpu_vs_mt.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log
echo finished
