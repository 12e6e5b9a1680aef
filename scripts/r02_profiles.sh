#!/bin/bash
# round 2: the measurement record of the shipped kernels (bench lines, rocprofv3 kernel stats, PMC passes)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02prof
mkdir -p $OUT
cd $ROOT
python bench.py --pipelined-extra > $OUT/bench_stepcloud.json 2> $OUT/bench_stepcloud.err; echo "bench step rc=$?"
python bench.py --workload landsatLike128 --steps 5 --warmup 1 > $OUT/bench_landsat.json 2> $OUT/bench_landsat.err; echo "bench landsat rc=$?"
python bench.py --block-walk 0 --no-cpu-baseline --no-secondary > $OUT/bench_stepcloud_facebyface.json 2> $OUT/bench_fbf.err; echo "bench fbf rc=$?"
BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-secondary --steps 5 > $OUT/bench_stepcloud_rccl1.json 2> $OUT/bench_rccl1.err; echo "bench rccl rc=$?"
python scripts/config4_bench.py > $OUT/config4.log 2>&1
python scripts/quick_bench.py --case radar --ppb 1000000 --batches 100 --thr 0 --reps 3 > $OUT/radar.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_step -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --event-threshold 16 > $OUT/stats_step.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_landsat -- python3 $ROOT/bench.py --workload landsatLike128 --steps 5 --warmup 1 --no-cpu-baseline --event-threshold 20 > $OUT/stats_landsat.log 2>&1
cd $ROOT
bash scripts/pmc_profile.sh step r02prof/pmc_step --thr 16 > $OUT/pmc_step.log 2>&1
bash scripts/pmc_profile.sh landsat r02prof/pmc_landsat --thr 20 > $OUT/pmc_landsat.log 2>&1
find $OUT -name "*kernel_stats.csv" | xargs -n1 head -4
echo finished
