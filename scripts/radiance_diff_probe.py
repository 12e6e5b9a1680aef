"""Development probe: one seed of test_random_domains_radiance_against_the_oracle -- the product as the test runs it, the product with
photons and rays that stop at every face, the oracle: direction means, and where in the field the difference of direction d sits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcbrat3d_amd as M
from tests import cases
from tests.test_gpu_intensity import random_radiance_case, SEED
from mcbrat3d_amd.integrator import new_RandomNumberSequence
from oracle import oracle as O
seed = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
case, rr, mus, phis, mu0, priv = random_radiance_case(seed)
nx, ny, nz = len(case["xe"]) - 1, len(case["ye"]) - 1, len(case["ze"]) - 1
print("seed", seed, "grid", nx, ny, nz, "rr", rr, "mus", mus, "phis", phis, "mu0", mu0, "priv", priv, "albedo", case["albedo"], "nc", len(case["components"]), flush=True)
def run(tuning):
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=9001, minForwardTableSize=9001, intensityMus=mus, intensityPhis=phis,
                            computeIntensity=True, useRussianRouletteForIntensity=rr, zetaMin=0.3)
    integ.setTuning(eventThreshold=24, **tuning)
    photons = M.new_PhotonStream(mu0, 40.0, numberOfPhotons=10 ** 9)
    integ.resetMoments()
    integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, n)
    res = integ.reportResults()
    integ.finalize()
    return res
a = run(dict(privateTallies=priv))
b = run(dict(privateTallies=0, layerSkip=0, blockWalk=0, brickLayout=0))
P = cases.oracle_problem(case, nsteps=9001)
I = cases.oracle_intensity(case, mus, phis, n_angles=9001, use_russian_roulette=rr, zeta_min=0.3)
ref = O.compute_radiative_transfer_intensity(P, O.solar_source(mu0, 40.0), O.philox_rng(SEED, 0), n, I)
r = ref["intensity"].reshape(-1, ny, nx).transpose(2, 1, 0)
print("means  test settings", a["meanIntensity"], "\n       face by face ", b["meanIntensity"], "\n       oracle       ", ref["meanIntensity"])
for name, g in (("test settings", a["intensity"]), ("face by face", b["intensity"])):
    for d in range(len(mus)):
        diff = g[:, :, d] - r[:, :, d]
        i = np.unravel_index(np.argmax(np.abs(diff)), diff.shape)
        print("  %s d=%d: sum of pixel differences %.6g (mean x %d pixels), largest %.6g at pixel %s (field mean %.4g); pixels differing by > 1e-6: %d"
              % (name, d, diff.sum(), nx * ny, diff[i], i, r[:, :, d].mean(), int((np.abs(diff) > 1e-6).sum())))
