#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c41
mkdir -p $OUT
cd $ROOT
for jt in 4 6 8 12; do
  echo "jt=$jt" >> $OUT/ab.log
  MCBRAT_JUMP_THRESHOLD=$jt timeout -k 10 400 python scripts/quick_bench.py --case landsat --ppb 1000000 --batches 100 --thr 20 --sthr 8 12 16 --lthr 4 8 12 --reps 2 >> $OUT/ab.log 2>&1 || exit 1
done
grep -E "jt=|case=" $OUT/ab.log | sed -e 's/bpc=0 priv=-1 block=-1//' -e 's/brick=-1 inflight=-1 ppb=1000000 nb=100//'
echo finished
