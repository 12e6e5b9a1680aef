"""Development probe: one seed of tests/test_gpu_intensity.py::test_random_domains_radiance_against_the_oracle, photons [first, first + n)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcbrat3d_amd as M
from tests import cases
from tests.test_gpu_intensity import random_radiance_case, SEED
from mcbrat3d_amd.integrator import new_RandomNumberSequence
seed = int(sys.argv[1]); n = int(sys.argv[2]); first = int(sys.argv[3])
case, rr, mus, phis, mu0, priv = random_radiance_case(seed)
dom = cases.product_domain(case)
integ = M.new_Integrator(dom)
integ.specifyParameters(minInverseTableSize=9001, minForwardTableSize=9001, intensityMus=mus, intensityPhis=phis,
                        computeIntensity=True, useRussianRouletteForIntensity=rr, zetaMin=0.3)
integ.setTuning(eventThreshold=24, privateTallies=priv)
photons = M.new_PhotonStream(mu0, 40.0, numberOfPhotons=10 ** 9)
integ.resetMoments()
r = new_RandomNumberSequence(SEED); r.nextPhotonId = first
print("seed %d photons %d..%d rr %s mus %s priv %d" % (seed, first, first + n, rr, mus, priv), flush=True)
integ.computeRadiativeTransfer(dom, r, photons, n)
print("  returned", integ.reportResults()["meanIntensity"], flush=True)
