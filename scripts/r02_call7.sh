#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c7
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/tests.log 2>&1; echo "tests rc=$?"
tail -5 $OUT/tests.log
timeout -k 10 300 python scripts/quick_bench.py --case landsat --ppb 1000000 --batches 100 --thr 24 32 40 --sthr 8 12 16 --reps 2 > $OUT/landsat_ab.log 2>&1
timeout -k 10 300 python scripts/quick_bench.py --case radar --ppb 1000000 --batches 100 --thr 32 --sthr 8 12 --reps 2 >> $OUT/landsat_ab.log 2>&1
timeout -k 10 300 python scripts/quick_bench.py --case step --bw 0 --thr 16 --sthr 8 12 --reps 3 >> $OUT/landsat_ab.log 2>&1
grep "case=" $OUT/landsat_ab.log | awk '{for(i=1;i<=NF;i++){if($i ~ /^case=|^thr=|^lthr=|^sthr=|^bw=/)printf "%s ",$i; if($i=="wall")printf "wall %s ",$(i+1)} print ""}'
echo finished
