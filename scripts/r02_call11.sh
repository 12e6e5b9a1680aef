#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c11
mkdir -p $OUT
cd $ROOT
for c in landsat radar; do
python scripts/ab_compare.py $c 32 >> $OUT/ab.log 2>&1
MCBRAT_LIB=$ROOT/ab/libmcbrat_dlj.so python scripts/ab_compare.py $c 32 >> $OUT/ab.log 2>&1
MCBRAT_LIB=$ROOT/ab/libmcbrat_dlj.so MCBRAT_JUMP_THRESHOLD=4 python scripts/ab_compare.py $c 32 >> $OUT/ab.log 2>&1
MCBRAT_LIB=$ROOT/ab/libmcbrat_dlj.so MCBRAT_JUMP_THRESHOLD=12 python scripts/ab_compare.py $c 32 >> $OUT/ab.log 2>&1
done
grep lib= $OUT/ab.log
BENCH_REHEARSE=1 timeout -k 10 600 python bench.py --gpus 2 --steps 3 --warmup 1 > $OUT/bench_rehearse2.json 2> $OUT/bench_rehearse2.err; echo "rehearse rc=$?"
tail -c 1500 $OUT/bench_rehearse2.json
BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > $OUT/bench_rccl1.json 2> $OUT/bench_rccl1.err; echo "rccl1 rc=$?"
python scripts/inten_bench.py > $OUT/inten.log 2>&1; tail -5 $OUT/inten.log
echo finished
