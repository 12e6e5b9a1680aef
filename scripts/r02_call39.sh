#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c39
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
bash scripts/pmc_profile.sh landsat r02c39/pmc_landsat --thr 20 > $OUT/pmc_landsat.log 2>&1
tail -40 $OUT/pmc_landsat.log
echo finished
