#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c10
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/tests.log 2>&1; echo "tests rc=$?"
tail -4 $OUT/tests.log
for c in landsat radar; do
python scripts/ab_compare.py $c 32 >> $OUT/ab.log 2>&1
for w in 7 8; do
MCBRAT_LIB=$ROOT/ab/libmcbrat_w$w.so python scripts/ab_compare.py $c 32 >> $OUT/ab.log 2>&1
done
done
MCBRAT_LIB=$ROOT/ab/libmcbrat_w7.so python scripts/ab_compare.py landsat 40 >> $OUT/ab.log 2>&1
MCBRAT_LIB=$ROOT/ab/libmcbrat_w7.so python scripts/ab_compare.py landsat 24 >> $OUT/ab.log 2>&1
grep lib= $OUT/ab.log
timeout -k 10 300 python scripts/quick_bench.py --case step --bw 1 0 --thr 16 --reps 3 > $OUT/step.log 2>&1; grep case= $OUT/step.log | awk '{for(i=1;i<=NF;i++){if($i ~ /^bw=/)printf "%s ",$i; if($i=="wall")printf "wall %s ",$(i+1)} print ""}'
echo finished
