#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c81
mkdir -p $OUT
cd $ROOT
MCBRAT_FLIGHT_FUZZ=${1:-40} timeout -k 10 1100 python -m pytest tests/test_gpu_flight.py --maxfail 10 -v -m gpu -k "${2:-midsize}" --timeout 120 --timeout-method thread > $OUT/tests.log 2>&1; rc=$?
grep -c PASSED $OUT/tests.log; grep -n "FAILED\|Timeout\|^E   *Assert\|^E  " $OUT/tests.log | head -30 | cut -c1-400; tail -2 $OUT/tests.log
echo "tests rc=$rc"
