#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c57
mkdir -p $OUT
cd $ROOT
for c in radar landsat overcast; do
timeout -k 10 400 python scripts/quick_bench.py --case $c --ppb 1000000 --batches 100 --thr 0 --reps 3 >> $OUT/tune.log 2>&1 || exit 1
done
grep -E "case=|chosen" $OUT/tune.log | sed -e 's/bpc=0 priv=-1 block=-1//' -e 's/lthr=0 sthr=0 brick=-1 inflight=-1 ppb=1000000 nb=100//'
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
python bench.py --workload radarLike128 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_radar.json 2> $OUT/bench_radar.err; echo "bench radar rc=$?"
python bench.py > $OUT/bench_step.json 2> $OUT/bench_step.err; echo "bench step rc=$?"
echo finished
