#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c20
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python scripts/quick_bench.py --case landsat --ppb 1000000 --batches 100 --thr 32 --reps 2 --counters --ssa 0.99 > $OUT/atomics.log 2>&1
timeout -k 10 300 python scripts/quick_bench.py --case landsat --ppb 1000000 --batches 100 --thr 32 --reps 2 --counters --ssa 1.0 >> $OUT/atomics.log 2>&1
grep -E "case=|per photon" $OUT/atomics.log
echo finished
