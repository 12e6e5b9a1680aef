#!/bin/bash
# clear-air flight: event threshold x jump threshold
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c31
mkdir -p $OUT
cd $ROOT
for jt in 4 8 16; do
  echo "jt=$jt" >> $OUT/sweep.log
  MCBRAT_JUMP_THRESHOLD=$jt timeout -k 10 300 python scripts/quick_bench.py --case landsat --ppb 1000000 --batches 100 --thr 24 32 40 48 --reps 2 >> $OUT/sweep.log 2>&1 || exit 1
done
MCBRAT_JUMP_THRESHOLD=8 timeout -k 10 300 python scripts/quick_bench.py --case radar --ppb 1000000 --batches 100 --thr 24 32 40 48 --reps 2 >> $OUT/sweep.log 2>&1 || exit 1
grep -E "jt=|case=" $OUT/sweep.log | sed -e 's/bpc=0 priv=-1 block=-1//' -e 's/lthr=0 sthr=0 brick=-1 inflight=-1 ppb=1000000 nb=100//'
echo finished
