"""Development probe: one seed of tests/test_gpu_parity.py::test_random_domains_against_the_oracle, face-by-face kernel against the
oracle: which fields of the per-photon fates differ, and for which kinds of photons."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcbrat3d_amd as M
from tests import cases
from tests.test_gpu_parity import random_oracle_case, SEED
from oracle import oracle as O
from mcbrat3d_amd.integrator import new_RandomNumberSequence
seed = int(sys.argv[1]); n = 15000
case, mu0, phi0, rr = random_oracle_case(seed)
P = cases.oracle_problem(case, nsteps=9001, use_russian_roulette=rr)
ref = O.compute_rt(P, O.solar_source(mu0, phi0), O.philox_rng(SEED, 0), n, want_fates=True)
rf = ref["fates"]
dom = cases.product_domain(case)
integ = M.new_Integrator(dom)
integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=rr)
integ.setTuning(eventThreshold=16, privateTallies=0, layerSkip=0)
photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 9)
got = integ.traceFates(dom, new_RandomNumberSequence(SEED), photons, n)
print("seed", seed, "grid", len(case["xe"]) - 1, len(case["ye"]) - 1, len(case["ze"]) - 1, "mu0 %.3f phi0 %.1f rr %s albedo %g nc %d" % (mu0, phi0, rr, case["albedo"], len(case["components"])))
for f in ("fate", "ix", "iy", "iz", "nScatter"):
    print("  differ in %-9s %.4f" % (f, (got[f] != rf[f]).mean()))
print("  differ in weight    %.4f" % (np.abs(got["weight"] - rf["weight"]) > 1e-6).mean())
d = np.nonzero((got["nScatter"] != rf["nScatter"]) | (got["fate"] != rf["fate"]) | (got["ix"] != rf["ix"]) | (got["iz"] != rf["iz"]))[0]
print("  first differing photons (kernel | oracle):")
for i in d[:10]:
    print("   ", i, got[i], "|", rf[i])
o = rf["nScatter"]
for lo, hi in ((0, 0), (1, 2), (3, 10), (11, 10 ** 6)):
    m = (o >= lo) & (o <= hi)
    if m.any():
        same = (got["fate"] == rf["fate"]) & (got["ix"] == rf["ix"]) & (got["iy"] == rf["iy"]) & (got["iz"] == rf["iz"]) & (got["nScatter"] == rf["nScatter"])
        print("  order %d..%d: %d photons, identical %.4f" % (lo, hi, m.sum(), same[m].mean()))
