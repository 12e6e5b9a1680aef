#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c79
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MCBRAT_FLIGHT_FUZZ=${1:-300}
cd $ROOT
timeout -k 10 1000 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 -m pytest tests/test_gpu_tunings.py -q -m gpu --maxfail 10 --timeout 120 --timeout-method thread > $OUT/tests.log 2>&1; rc=$?
tail -3 $OUT/tests.log
find $OUT/prof -name "*kernel_stats.csv" | xargs -n1 cat | cut -d, -f1-3 | cut -c1-150
echo "rc=$rc"
