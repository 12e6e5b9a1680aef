#!/bin/bash
# soak run of every random differential test (seeds 0 .. N-1), one file after the other; a file stops after 20 failures,
# the run at the first hang (a test killed by its time limit: no further GPU step after that)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02soak
mkdir -p $OUT
cd $ROOT
N=${1:-1000}
for f in test_gpu_flight test_gpu_layer_skip test_gpu_block_walk test_gpu_parity test_gpu_intensity test_gpu_tunings; do
  MCBRAT_FLIGHT_FUZZ=$N timeout -k 10 1100 python -m pytest tests/$f.py --maxfail 20 -v -m gpu -k "random" --timeout 120 --timeout-method thread > $OUT/$f.log 2>&1; rc=$?
  echo "$f rc=$rc passed=$(grep -c PASSED $OUT/$f.log) : $(tail -1 $OUT/$f.log)"
  grep -n "FAILED\|Timeout\|^E   *Assert" $OUT/$f.log | head -5 | cut -c1-300
  if grep -q "Timeout" $OUT/$f.log || [ $rc -ge 124 ]; then echo "stopping at $f"; exit 1; fi
done
echo finished
