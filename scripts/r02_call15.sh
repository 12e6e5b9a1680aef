#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c15
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_block_walk.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_vs_mt.py -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?"
tail -4 $OUT/tests.log
timeout -k 10 300 python scripts/quick_bench.py --case step --bw 1 --thr 12 16 24 --reps 3 --counters > $OUT/step.log 2>&1; grep case= $OUT/step.log | awk '{for(i=1;i<=NF;i++){if($i ~ /^bw=|^thr=/)printf "%s ",$i; if($i=="wall")printf "wall %s ",$(i+1); if($i=="kernel")printf "kernel %s ",$(i+1)} print ""}'
timeout -k 10 300 python scripts/quick_bench.py --case plane --bw 1 --thr 16 --reps 3 >> $OUT/step.log 2>&1; grep "case=plane" $OUT/step.log | awk '{for(i=1;i<=NF;i++){if($i=="wall")printf "plane wall %s ",$(i+1)} print ""}'
echo finished
