#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c44
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_intensity.py tests/test_gpu_layer_skip.py -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $OUT/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python scripts/inten_landsat.py 20 > $OUT/inten.log 2>&1; cat $OUT/inten.log
echo finished
