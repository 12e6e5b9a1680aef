#!/bin/bash
# round 2, GPU call 1: tests of the refactors, baseline numbers of the round-1 kernels, domain-size experiment, counters
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c1
mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/tests.log
for n in 32 64 128; do
  python scripts/quick_bench.py --case landsat --n $n --ppb 1000000 --batches 100 --thr 32 --reps 2 --counters >> $OUT/size_sweep.log 2>&1
done
tail -12 $OUT/size_sweep.log
python bench.py > $OUT/bench_step.json 2> $OUT/bench_step.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_step -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --event-threshold 16 > $OUT/stats_step.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_landsat -- python3 $ROOT/bench.py --workload landsatLike128 --steps 5 --warmup 1 --no-cpu-baseline --event-threshold 32 > $OUT/stats_landsat.log 2>&1
cd $ROOT
bash scripts/pmc_profile.sh step r02c1/pmc_step --thr 16 > $OUT/pmc_step.log 2>&1
bash scripts/pmc_profile.sh landsat r02c1/pmc_landsat --thr 32 > $OUT/pmc_landsat.log 2>&1
find $OUT -name "*kernel_stats.csv" | head
echo finished
