#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
MCBRAT_TRACE_PHOTON=3 timeout -k 5 100 python - <<'PY' 2>&1 | grep -v "^  differ\|order" | tail -30
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
import mcbrat3d_amd as M
from tests import cases
from tests.test_gpu_parity import random_oracle_case, SEED
from mcbrat3d_amd.integrator import new_RandomNumberSequence
case, mu0, phi0, rr = random_oracle_case(7)
dom = cases.product_domain(case)
integ = M.new_Integrator(dom)
integ.specifyParameters(minInverseTableSize=2001, useRayTracing=True, useRussianRoulette=rr)
integ.setTuning(eventThreshold=16, privateTallies=0, layerSkip=0)
photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 9)
got = integ.traceFates(dom, new_RandomNumberSequence(SEED), photons, 8)
print(got[3])
PY
