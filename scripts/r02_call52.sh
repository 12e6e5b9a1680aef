#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c52
mkdir -p $OUT
cd $ROOT
for h in 5 10 20 40; do
  echo "haze x$h" >> $OUT/haze.log
  timeout -k 10 300 python scripts/quick_bench.py --case hazy --haze $h --ppb 1000000 --batches 100 --thr 20 --skip 2 3 --reps 2 >> $OUT/haze.log 2>&1 || exit 1
done
grep -E "haze|case=" $OUT/haze.log | sed -e 's/bpc=0 priv=-1 block=-1//' -e 's/lthr=0 sthr=0 brick=-1 inflight=-1 ppb=1000000 nb=100//'
echo finished
