#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c66
mkdir -p $OUT; rm -f $OUT/ab.log
cd $ROOT
for rep in 1 2 3; do
timeout -k 10 200 python scripts/quick_bench.py --case step --ppb 100000 --batches 100 --thr 16 --reps 5 >> $OUT/ab.log 2>&1 || exit 1
echo "   ^ this tree" >> $OUT/ab.log
MCBRAT_LIB=$ROOT/ab/libmcbrat_prev.so timeout -k 10 200 python scripts/quick_bench.py --case step --ppb 100000 --batches 100 --thr 16 --reps 5 >> $OUT/ab.log 2>&1 || exit 1
echo "   ^ block walk before the clamps" >> $OUT/ab.log
done
grep -A1 "case=" $OUT/ab.log | grep -v "^--" | sed -e 's/bpc=0 priv=-1 block=-1//' -e 's/lthr=0 sthr=0 brick=-1 inflight=-1 ppb=100000 nb=100//'
echo finished
