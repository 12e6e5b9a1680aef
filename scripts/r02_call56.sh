#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c56
mkdir -p $OUT
cd $ROOT
for tp in 16777216 33554432 67108864 134217728; do
for c in radar landsat; do
echo "tune photons $tp case $c" >> $OUT/tune.log
MCBRAT_TUNE_PHOTONS=$tp timeout -k 10 400 python scripts/quick_bench.py --case $c --ppb 1000000 --batches 200 --thr 0 --reps 1 >> $OUT/tune.log 2>&1 || exit 1
MCBRAT_TUNE_PHOTONS=$tp timeout -k 10 400 python scripts/quick_bench.py --case $c --ppb 1000000 --batches 200 --thr 0 --reps 1 >> $OUT/tune.log 2>&1 || exit 1
done
done
grep -E "tune photons|chosen" $OUT/tune.log
echo finished
