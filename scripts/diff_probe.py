"""First photons whose fate differs between the product and the oracle (same Philox streams), and -- with a photon index as the
second argument -- both sides' per-collision records of that photon (MCBRAT_TRACE_PHOTON / ORC_TRACE_PHOTON, on stderr).
usage: python scripts/diff_probe.py [n photons] [photon index]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from tests import cases  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
if len(sys.argv) > 2:
    os.environ["MCBRAT_TRACE_PHOTON"] = os.environ["ORC_TRACE_PHOTON"] = sys.argv[2]
import mcbrat3d_amd as M  # noqa: E402
from mcbrat3d_amd.integrator import new_RandomNumberSequence  # noqa: E402
from oracle import oracle as O  # noqa: E402

case = cases.landsat_like(n=48, nz=24, n_entries=6)
dom = cases.product_domain(case)
integ = M.new_Integrator(dom)
integ.specifyParameters(minInverseTableSize=9001)
integ.setTuning(eventThreshold=16)
photons = M.new_PhotonStream(0.5, 30.0, numberOfPhotons=10 ** 9)
got = integ.traceFates(dom, new_RandomNumberSequence(90210), photons, n)
print("walk mode", integ.walkMode(), "bad", integ.badPhotons(), flush=True)
P = cases.oracle_problem(case, nsteps=9001)
rf = O.compute_rt(P, O.solar_source(0.5, 30.0), O.philox_rng(90210, 0), n, want_fates=True)["fates"]
same = (got["fate"] == rf["fate"]) & (got["ix"] == rf["ix"]) & (got["iy"] == rf["iy"]) & (got["nScatter"] == rf["nScatter"])
print("identical %.4f; mean scatterings gpu %.3f oracle %.3f; fates gpu %s oracle %s" % (
    same.mean(), got["nScatter"].mean(), rf["nScatter"].mean(), np.bincount(got["fate"], minlength=4), np.bincount(rf["fate"], minlength=4)))
bad = np.flatnonzero(~same)[:12]
for i in bad:
    print(i, "gpu", got[i], "oracle", rf[i])
integ.finalize()
