#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_step -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --event-threshold 16 > $OUT/stats_step.log 2>&1
cd $ROOT
bash scripts/pmc_profile.sh step r02c4/pmc_step --thr 16 > $OUT/pmc_step.log 2>&1
tail -40 $OUT/pmc_step.log
find $OUT -name "*kernel_stats.csv" | xargs head -5
echo finished
