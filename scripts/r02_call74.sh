#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c74
mkdir -p $OUT
cd $ROOT
FIRST=$1; COUNT=$2; CHUNK=$3
for ((f=FIRST; f<FIRST+COUNT; f+=CHUNK)); do
  timeout -k 5 30 python scripts/radiance_probe.py 22 $CHUNK $f > $OUT/chunk_$f.log 2>&1
  rc=$?
  echo "chunk $f +$CHUNK rc=$rc $(tail -1 $OUT/chunk_$f.log | cut -c1-100)"
  if [ $rc -ne 0 ]; then echo "HUNG in [$f, $((f+CHUNK)))"; break; fi
done
echo finished
