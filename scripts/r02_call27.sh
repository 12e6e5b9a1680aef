#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c27
mkdir -p $OUT
cd $ROOT
for c in landsat radar; do
python scripts/ab_compare.py $c 32 >> $OUT/ab.log 2>&1
MCBRAT_LIB=$ROOT/ab/libmcbrat_prec.so python scripts/ab_compare.py $c 32 >> $OUT/ab.log 2>&1
done
grep lib= $OUT/ab.log
echo finished
