"""Development probe: one seed of tests/test_gpu_parity.py::test_random_domains_against_the_oracle with ONE walk
(argv: seed walk[0 face by face | 1 layers + flight | 2 LDS face by face | 3 block walk] [photons] [first])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcbrat3d_amd as M
from tests import cases
from tests.test_gpu_parity import random_oracle_case, SEED
from mcbrat3d_amd.integrator import new_RandomNumberSequence
seed, walk = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 15000
first = int(sys.argv[4]) if len(sys.argv) > 4 else 0
tuning = [dict(privateTallies=0, layerSkip=0), dict(privateTallies=0, layerSkip=3), dict(blockWalk=0), dict(blockWalk=2)][walk]
case, mu0, phi0, rr = random_oracle_case(seed)
dom = cases.product_domain(case)
integ = M.new_Integrator(dom)
integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=rr)
integ.setTuning(eventThreshold=16, **tuning)
photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 9)
print("seed %d walk %d: grid %d %d %d mu0 %g phi0 %.1f rr %s albedo %g; photons %d..%d" % (seed, walk, len(case["xe"]) - 1, len(case["ye"]) - 1, len(case["ze"]) - 1, mu0, phi0, rr, case["albedo"], first, first + n), flush=True)
r = new_RandomNumberSequence(SEED)
r.nextPhotonId = first
fates = integ.traceFates(dom, r, photons, n)
print("  returned:", integ.walkMode(), np.bincount(np.maximum(fates["fate"], 0), minlength=3), flush=True)
