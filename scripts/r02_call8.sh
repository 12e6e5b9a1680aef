#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c8
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/tests.log 2>&1; echo "tests rc=$?"
tail -5 $OUT/tests.log
for jt in 1 4 8 12 16; do
MCBRAT_JUMP_THRESHOLD=$jt timeout -k 10 300 python scripts/quick_bench.py --case landsat --ppb 1000000 --batches 100 --thr 32 --reps 2 | sed "s/^case/jt=$jt case/" >> $OUT/landsat_ab.log 2>&1
done
MCBRAT_JUMP_THRESHOLD=8 timeout -k 10 300 python scripts/quick_bench.py --case radar --ppb 1000000 --batches 100 --thr 32 --reps 2 | sed "s/^case/jt=8 case/" >> $OUT/landsat_ab.log 2>&1
grep "case=" $OUT/landsat_ab.log | awk '{for(i=1;i<=NF;i++){if($i ~ /^jt=|^case=|^thr=|^sthr=/)printf "%s ",$i; if($i=="wall")printf "wall %s ",$(i+1)} print ""}'
echo finished
