#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c73
mkdir -p $OUT
cd $ROOT
MCBRAT_FLIGHT_FUZZ=${1:-8} timeout -k 10 1000 python -m pytest tests/test_gpu_intensity.py -v -m gpu -k "radiance_against_the_oracle" --timeout 200 --timeout-method thread > $OUT/tests.log 2>&1; rc=$?
grep -c PASSED $OUT/tests.log; grep -n "FAILED\|Timeout\|^E   *Assert" $OUT/tests.log | head -20 | cut -c1-300; tail -2 $OUT/tests.log
echo "tests rc=$rc"
