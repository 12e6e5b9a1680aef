#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c12
mkdir -p $OUT
cd $ROOT
MCBRAT_LIB=$ROOT/ab/libmcbrat_stamps.so python scripts/quick_bench.py --case landsat --ppb 1000000 --batches 20 --thr 32 --reps 1 --counters > $OUT/stamps_landsat.log 2>&1
MCBRAT_LIB=$ROOT/ab/libmcbrat_stamps.so python scripts/quick_bench.py --case radar --ppb 1000000 --batches 20 --thr 32 --reps 1 --counters > $OUT/stamps_radar.log 2>&1
grep -E "stamp|walk iters|per photon" $OUT/stamps_landsat.log $OUT/stamps_radar.log
python scripts/ab_compare.py landsat 32
echo finished
