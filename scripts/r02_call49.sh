#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c49
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $OUT/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python scripts/quick_bench.py --case hazy --ppb 1000000 --batches 100 --thr 20 --skip 2 1 3 --reps 2 > $OUT/hazy.log 2>&1 || exit 1
grep -E "case=" $OUT/hazy.log | sed -e 's/bpc=0 priv=-1 block=-1//' -e 's/lthr=0 sthr=0 brick=-1 inflight=-1 ppb=1000000 nb=100//'
bash scripts/ab_prev.sh 20 | grep lib=
echo finished
