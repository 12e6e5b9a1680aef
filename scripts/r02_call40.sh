#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c40
mkdir -p $OUT
cd $ROOT
for c in landsat radar; do
for thr in 16 24; do
timeout -k 10 200 python scripts/ab_compare.py $c $thr >> $OUT/ab.log 2>&1 || exit 1
for w in 4 5 7; do
MCBRAT_LIB=$ROOT/ab/libmcbrat_w$w.so timeout -k 10 200 python scripts/ab_compare.py $c $thr >> $OUT/ab.log 2>&1 || exit 1
done
done
done
grep lib= $OUT/ab.log
echo finished
