#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c54
mkdir -p $OUT
cd $ROOT
MCBRAT_FLIGHT_FUZZ=300 timeout -k 10 1000 python -m pytest tests/test_gpu_flight.py -q -m gpu -k "random" > $OUT/tests.log 2>&1; rc=$?
tail -30 $OUT/tests.log
echo "tests rc=$rc"
