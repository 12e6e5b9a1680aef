#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c21
mkdir -p $OUT
cd $ROOT
for c in landsat radar; do
MCBRAT_XCD_TALLIES=0 python scripts/ab_compare.py $c 32 | sed 's/^lib=/xcd=0 lib=/' >> $OUT/ab.log 2>&1
MCBRAT_XCD_TALLIES=1 python scripts/ab_compare.py $c 32 | sed 's/^lib=/xcd=1 lib=/' >> $OUT/ab.log 2>&1
MCBRAT_XCD_TALLIES=1 python scripts/ab_compare.py $c 24 | sed 's/^lib=/xcd=1 lib=/' >> $OUT/ab.log 2>&1
done
cat $OUT/ab.log | grep -E "lib=|Error|error" | head -20
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/tests.log 2>&1; echo "tests rc=$?"
tail -5 $OUT/tests.log
echo finished
