#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c58
mkdir -p $OUT
cd $ROOT
MCBRAT_FLIGHT_FUZZ=400 timeout -k 10 1000 python -m pytest tests/test_gpu_layer_skip.py -q -m gpu -k "random" > $OUT/tests.log 2>&1; rc=$?
tail -15 $OUT/tests.log
echo "tests rc=$rc"
