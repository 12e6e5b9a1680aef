#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c38
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?
tail -5 $OUT/tests.log
echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
python bench.py --workload landsatLike128 --steps 5 --warmup 1 > $OUT/bench_landsat.json 2> $OUT/bench_landsat.err; echo "bench landsat rc=$?"
python bench.py > $OUT/bench_stepcloud.json 2> $OUT/bench_stepcloud.err; echo "bench step rc=$?"
timeout -k 10 300 python scripts/quick_bench.py --case radar --ppb 1000000 --batches 100 --thr 0 --reps 3 > $OUT/radar.log 2>&1; grep case= $OUT/radar.log
echo finished
