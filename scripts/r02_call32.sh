#!/bin/bash
# clear-air flight: with and without absorption tallies (omega0 = 1: no volume atomics)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c32
mkdir -p $OUT
cd $ROOT
for ssa in 1.0 0.99; do
  timeout -k 10 300 python scripts/quick_bench.py --case landsat --ssa $ssa --ppb 1000000 --batches 100 --thr 32 --skip 2 1 --reps 2 --counters >> $OUT/ssa.log 2>&1 || exit 1
done
grep -E "case=|per photon|walk iters" $OUT/ssa.log | sed -e 's/bpc=0 priv=-1 block=-1//' -e 's/lthr=0 sthr=0 brick=-1 inflight=-1 ppb=1000000 nb=100//'
echo finished
