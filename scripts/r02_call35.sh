#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c35
mkdir -p $OUT
cd $ROOT
for jt in 4 6 8; do
  echo "split jt=$jt" >> $OUT/ab.log
  MCBRAT_JUMP_THRESHOLD=$jt MCBRAT_LIB=$ROOT/ab/libmcbrat_split.so timeout -k 10 300 python scripts/quick_bench.py --case landsat --ppb 1000000 --batches 100 --thr 32 --reps 2 >> $OUT/ab.log 2>&1 || exit 1
done
echo "default jt=8" >> $OUT/ab.log
timeout -k 10 300 python scripts/quick_bench.py --case landsat --ppb 1000000 --batches 100 --thr 32 --reps 2 >> $OUT/ab.log 2>&1 || exit 1
grep -E "jt=|case=" $OUT/ab.log | sed -e 's/bpc=0 priv=-1 block=-1//' -e 's/lthr=0 sthr=0 brick=-1 inflight=-1 ppb=1000000 nb=100//'
echo finished
