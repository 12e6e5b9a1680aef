#!/bin/bash
# the bench lines of profiles/ (run after profiles/pmc_shipped.json is in place, so that roofline.valu is built from it)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02lines
mkdir -p $OUT
cd $ROOT
python bench.py --pipelined-extra > $OUT/bench_stepcloud.json 2> $OUT/bench_stepcloud.err; echo "bench step rc=$?"
python bench.py --workload landsatLike128 --steps 5 --warmup 1 > $OUT/bench_landsat.json 2> $OUT/bench_landsat.err; echo "bench landsat rc=$?"
python bench.py --block-walk 0 --no-cpu-baseline --no-secondary > $OUT/bench_stepcloud_facebyface.json 2> $OUT/bench_fbf.err; echo "bench fbf rc=$?"
BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-secondary --steps 5 > $OUT/bench_stepcloud_rccl1.json 2> $OUT/bench_rccl1.err; echo "bench rccl rc=$?"
echo finished
