#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c50
mkdir -p $OUT
cd $ROOT
MCBRAT_LIB=$ROOT/ab/libmcbrat_b1024.so timeout -k 10 300 python scripts/quick_bench.py --case step --ppb 100000 --batches 100 --thr 16 --block 768 1024 768 1024 --reps 3 > $OUT/step.log 2>&1 || { tail -5 $OUT/step.log; exit 1; }
grep case= $OUT/step.log | sed -e 's/lthr=0 sthr=0 brick=-1 inflight=-1//'
echo finished
