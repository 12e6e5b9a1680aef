#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c19
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python scripts/quick_bench.py --case step --bw 1 --thr 16 --reps 3 --batches 100 > $OUT/drain.log 2>&1
timeout -k 10 300 python scripts/quick_bench.py --case step --bw 1 --thr 16 --reps 3 --batches 400 >> $OUT/drain.log 2>&1
timeout -k 10 300 python scripts/quick_bench.py --case step --bw 1 --thr 16 --reps 3 --batches 2000 >> $OUT/drain.log 2>&1
timeout -k 10 300 python scripts/quick_bench.py --case step --bw 1 --thr 16 --reps 3 --batches 100 --ppb 1000000 >> $OUT/drain.log 2>&1
grep case= $OUT/drain.log | awk '{for(i=1;i<=NF;i++){if($i ~ /^ppb=|^nb=/)printf "%s ",$i; if($i=="wall")printf "wall %s ",$(i+1); if($i=="kernel")printf "kernel %s ",$(i+1)} print ""}'
echo finished
