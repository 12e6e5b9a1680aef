#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c55
mkdir -p $OUT
cd $ROOT
for c in radar landsat; do
timeout -k 10 400 python scripts/quick_bench.py --case $c --ppb 1000000 --batches 100 --thr 0 16 20 24 0 --reps 3 >> $OUT/tune.log 2>&1 || exit 1
done
grep -E "case=|chosen" $OUT/tune.log | sed -e 's/bpc=0 priv=-1 block=-1//' -e 's/lthr=0 sthr=0 brick=-1 inflight=-1 ppb=1000000 nb=100//'
echo finished
