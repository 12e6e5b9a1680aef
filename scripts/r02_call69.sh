#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
echo "== default"; timeout -k 5 100 python scripts/oracle_probe.py 7 2>&1 | grep "order\|differ in ix"
echo "== MCBRAT_REGULAR_WALK=0"; MCBRAT_REGULAR_WALK=0 timeout -k 5 100 python scripts/oracle_probe.py 7 2>&1 | grep "order\|differ in ix"
