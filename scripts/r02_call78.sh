#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c78
mkdir -p $OUT
cd $ROOT
MCBRAT_FLIGHT_FUZZ=${1:-60} timeout -k 10 1000 python -m pytest tests/test_gpu_tunings.py --maxfail 10 -v -m gpu -k "${2:-random}" --timeout 120 --timeout-method thread > $OUT/tests.log 2>&1; rc=$?
grep -c PASSED $OUT/tests.log; grep -n "FAILED\|Timeout\|^E   *Assert" $OUT/tests.log | head -20 | cut -c1-400; tail -2 $OUT/tests.log
echo "tests rc=$rc"
