#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c18
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?"
tail -4 $OUT/tests.log
python scripts/config4_bench.py > $OUT/config4.log 2>&1; cat $OUT/config4.log | head -4
python scripts/ab_compare.py landsat 32; python scripts/ab_compare.py radar 32
echo finished
