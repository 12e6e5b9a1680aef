#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c51
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python scripts/skip_bias_probe.py > $OUT/bias.log 2>&1; cat $OUT/bias.log
echo finished
