#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c75
mkdir -p $OUT
cd $ROOT
MCBRAT_LIB=$ROOT/ab/libmcbrat_ng.so MCBRAT_FLIGHT_FUZZ=60 timeout -k 10 400 python -m pytest tests/test_gpu_intensity.py -v -m gpu -k "radiance_against_the_oracle" --timeout 60 --timeout-method thread > $OUT/tests.log 2>&1; rc=$?
grep -c PASSED $OUT/tests.log; grep -n "FAILED\|Timeout" $OUT/tests.log | head -5 | cut -c1-200; tail -1 $OUT/tests.log
echo "tests rc=$rc"
