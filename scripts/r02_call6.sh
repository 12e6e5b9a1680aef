#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c6
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_block_walk.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?"
tail -5 $OUT/tests.log
timeout -k 10 300 python scripts/quick_bench.py --case step --bw 1 --block 768 --thr 12 16 20 --lthr 4 8 12 --sthr 6 8 12 16 --reps 3 > $OUT/step_ab.log 2>&1
for ct in 4 16; do MCBRAT_CROSS_THRESHOLD=$ct timeout -k 10 300 python scripts/quick_bench.py --case step --bw 1 --block 768 --thr 16 --reps 3 | sed "s/^case/ct=$ct case/" >> $OUT/step_ab.log 2>&1; done
grep "case=" $OUT/step_ab.log | sort -t' ' -k16 | awk '{print $6,$9,$10,$11,$17,$18,$19,$20,$21}' | sort -k6 -g | tail -40
echo finished
