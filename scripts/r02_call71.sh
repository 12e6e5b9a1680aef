#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout -k 5 60 python scripts/walk_probe.py 71 3 > gpurun_out/walk_3.log 2>&1; rc=$?; echo "seed 71 block walk rc=$rc"; tail -1 gpurun_out/walk_3.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
bash scripts/r02_call68.sh 500
