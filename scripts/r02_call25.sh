#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c25
mkdir -p $OUT
cd $ROOT
for c in landsat radar; do
for l in comb1w6 comb2w6; do
MCBRAT_LIB=$ROOT/ab/libmcbrat_$l.so python scripts/ab_compare.py $c 32 >> $OUT/ab.log 2>&1
done
done
grep lib= $OUT/ab.log
echo finished
