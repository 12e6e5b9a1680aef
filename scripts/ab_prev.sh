#!/bin/bash
# same-box A/B: the library in the tree against ab/libmcbrat_prev.so (built from another revision), interleaved
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab_prev
mkdir -p $OUT; rm -f $OUT/ab.log
cd $ROOT
THR=${1:-20}
for rep in 1 2; do
for c in landsat radar; do
timeout -k 10 200 python scripts/ab_compare.py $c $THR >> $OUT/ab.log 2>&1 || exit 1
MCBRAT_LIB=$ROOT/ab/libmcbrat_prev.so timeout -k 10 200 python scripts/ab_compare.py $c $THR >> $OUT/ab.log 2>&1 || exit 1
done
done
grep lib= $OUT/ab.log | sort -k2,2 -k1,1
echo finished
