#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c13
mkdir -p $OUT
cd $ROOT
for c in landsat radar; do
MCBRAT_TABLE_PAIRS=0 python scripts/ab_compare.py $c 32 | sed 's/^lib=/pairs=0 lib=/' >> $OUT/ab.log 2>&1
python scripts/ab_compare.py $c 32 | sed 's/^lib=/pairs=1 lib=/' >> $OUT/ab.log 2>&1
MCBRAT_LIB=$ROOT/ab/libmcbrat_prec7.so python scripts/ab_compare.py $c 32 | sed 's/^lib=/pairs=1 lib=/' >> $OUT/ab.log 2>&1
MCBRAT_LIB=$ROOT/ab/libmcbrat_prec6.so python scripts/ab_compare.py $c 32 | sed 's/^lib=/pairs=1 lib=/' >> $OUT/ab.log 2>&1
done
grep lib= $OUT/ab.log
echo finished
