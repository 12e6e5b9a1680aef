#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c34
mkdir -p $OUT
cd $ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?
tail -15 $OUT/tests.log
echo "tests rc=$rc"
exit $rc
