#!/bin/bash
# soak run of the random tests of ONE test file: r02_soak_one.sh <file without .py> <seeds> [-k expression]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02soak1
mkdir -p $OUT
cd $ROOT
f=$1
MCBRAT_FLIGHT_FUZZ=${2:-1000} timeout -k 10 1150 python -m pytest tests/$f.py --maxfail 20 -v -m gpu -k "${3:-random}" --timeout 120 --timeout-method thread > $OUT/$f.log 2>&1; rc=$?
echo "$f rc=$rc passed=$(grep -c PASSED $OUT/$f.log) : $(tail -1 $OUT/$f.log)"
grep -n "FAILED\|Timeout\|^E   *Assert" $OUT/$f.log | head -12 | cut -c1-300
