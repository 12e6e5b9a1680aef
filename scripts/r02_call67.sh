#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash scripts/r02_call66.sh 2>&1 | tail -8
bash scripts/r02_call60.sh 400 2>&1 | tail -5
