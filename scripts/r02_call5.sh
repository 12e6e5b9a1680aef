#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c5
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?"
tail -15 $OUT/tests.log
timeout -k 10 300 python scripts/quick_bench.py --case step --bw 1 --block 768 --thr 8 16 24 --reps 3 --counters > $OUT/step_ab.log 2>&1
grep "case=" $OUT/step_ab.log
timeout -k 10 300 python scripts/config4_bench.py > $OUT/config4.log 2>&1; cat $OUT/config4.log
echo finished
