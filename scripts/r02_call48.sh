#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c48
mkdir -p $OUT
cd $ROOT
echo "== previous (no request filters)" > $OUT/hazy.log
MCBRAT_LIB=$ROOT/ab/libmcbrat_prev.so timeout -k 10 400 python scripts/quick_bench.py --case hazy --ppb 1000000 --batches 100 --thr 20 --skip 2 1 --reps 2 --counters >> $OUT/hazy.log 2>&1 || exit 1
echo "== this tree" >> $OUT/hazy.log
timeout -k 10 400 python scripts/quick_bench.py --case hazy --ppb 1000000 --batches 100 --thr 20 --skip 2 1 --reps 2 --counters >> $OUT/hazy.log 2>&1 || exit 1
grep -E "==|case=|walk iters|per photon" $OUT/hazy.log | sed -e 's/bpc=0 priv=-1 block=-1//' -e 's/lthr=0 sthr=0 brick=-1 inflight=-1 ppb=1000000 nb=100//'
bash scripts/ab_prev.sh 20 | grep lib=
echo finished
