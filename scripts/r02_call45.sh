#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c45
mkdir -p $OUT
cd $ROOT
echo "== previous library" > $OUT/inten.log
MCBRAT_LIB=$ROOT/ab/libmcbrat_prev.so timeout -k 10 600 python scripts/inten_landsat.py 50 >> $OUT/inten.log 2>&1
echo "== this tree" >> $OUT/inten.log
timeout -k 10 600 python scripts/inten_landsat.py 50 >> $OUT/inten.log 2>&1
cat $OUT/inten.log
echo finished
