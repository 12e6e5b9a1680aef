#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c82
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MCBRAT_FLIGHT_FUZZ=${1:-100}
cd $ROOT
timeout -k 10 1000 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 -m pytest tests/test_gpu_flight.py -q -m gpu -k midsize --maxfail 10 --timeout 120 --timeout-method thread > $OUT/tests.log 2>&1; rc=$?
grep -n "passed\|failed" $OUT/tests.log | tail -2
find $OUT/prof -name "*kernel_stats.csv" | xargs -n1 cat | grep trace | cut -d, -f1-2 | cut -c1-150
echo "rc=$rc"
