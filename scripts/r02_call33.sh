#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c33
mkdir -p $OUT
cd $ROOT
for skip in 2 1; do
MCBRAT_LIB=$ROOT/ab/libmcbrat_stamps.so timeout -k 10 300 python scripts/quick_bench.py --case landsat --ppb 1000000 --batches 20 --thr 32 --reps 1 --skip $skip --counters >> $OUT/stamps_landsat.log 2>&1 || exit 1
done
grep -E "stamp|walk iters|per photon|case=" $OUT/stamps_landsat.log
echo finished
