#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c16
mkdir -p $OUT
cd $ROOT
for c in landsat radar; do
for t in 24 32 40; do python scripts/ab_compare.py $c $t >> $OUT/ab.log 2>&1; done
done
grep lib= $OUT/ab.log
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/tests.log 2>&1; echo "tests rc=$?"
tail -4 $OUT/tests.log
echo finished
