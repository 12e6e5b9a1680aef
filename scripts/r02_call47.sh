#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c47
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python scripts/quick_bench.py --case overcast --ppb 1000000 --batches 100 --thr 20 --skip 2 1 --reps 2 --counters > $OUT/overcast.log 2>&1 || exit 1
grep -E "case=|walk iters" $OUT/overcast.log | sed -e 's/bpc=0 priv=-1 block=-1//' -e 's/lthr=0 sthr=0 brick=-1 inflight=-1 ppb=1000000 nb=100//'
for c in landsat radar; do timeout -k 10 200 python scripts/ab_compare.py $c 20 >> $OUT/ab.log 2>&1 || exit 1; done
grep lib= $OUT/ab.log
timeout -k 10 600 python -m pytest tests/test_gpu_flight.py tests/test_gpu_layer_skip.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
echo finished
