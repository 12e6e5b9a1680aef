#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c37
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_flight.py -q -m gpu > $OUT/tests.log 2>&1; rc=$?
tail -40 $OUT/tests.log
echo "tests rc=$rc"
exit $rc
