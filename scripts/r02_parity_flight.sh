#!/bin/bash
# round 2: large-sample parity of the kernels with the clear-air flight against the oracle's MT mode
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02parity_flight
mkdir -p $OUT
cd $ROOT
python bench.py --workload landsatLike128 --steps 3 --warmup 1 --parity-photons 1000000000 --cpu-photons-per-core 20000000 > $OUT/parity_landsat.json 2> $OUT/parity_landsat.err; echo "landsat rc=$?"
python bench.py --workload radarLike128 --steps 3 --warmup 1 --parity-photons 400000000 --cpu-photons-per-core 16000000 > $OUT/parity_radar.json 2> $OUT/parity_radar.err; echo "radar rc=$?"
echo finished
