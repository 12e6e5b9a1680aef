"""Deeper parity check (development): fraction of photon histories identical to the oracle on large samples,
and where the differing ones come from."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import oracle as O  # noqa: E402
from tests import cases  # noqa: E402
import mcbrat3d_amd as M  # noqa: E402
from mcbrat3d_amd.integrator import new_RandomNumberSequence  # noqa: E402

for name, case, mu0, phi0, n in (("step", cases.step_cloud(0.99), 1.0, 0.0, 1000000),
                                 ("landsat", cases.landsat_like(), 0.5, 30.0, 200000)):
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=10001)
    photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 9)
    got = integ.traceFates(dom, new_RandomNumberSequence(4242), photons, n)
    P = cases.oracle_problem(case)
    t = time.time()
    ref = O.compute_rt(P, O.solar_source(mu0, phi0), O.philox_rng(4242, 0), n, want_fates=True)["fates"]
    same = (got["fate"] == ref["fate"]) & (got["ix"] == ref["ix"]) & (got["iy"] == ref["iy"]) & \
        (got["nScatter"] == ref["nScatter"]) & (np.abs(got["weight"] - ref["weight"]) <= 1e-6)
    bad = np.where(~same)[0]
    print(name, "n", n, "identical %.6f" % same.mean(), "differing", len(bad), "oracle %.1fs" % (time.time() - t))
    # net effect of the differing histories on the three domain fluxes
    for k, f in (("up", 0), ("down", 1)):
        dg = got["weight"][bad][got["fate"][bad] == f].sum()
        dr = ref["weight"][bad][ref["fate"][bad] == f].sum()
        print("   flux", k, "net difference / n = %.3e" % ((dg - dr) / n))
    print("   first differing:", [(int(i), tuple(int(got[q][i]) for q in ("fate", "ix", "nScatter")),
                                  tuple(int(ref[q][i]) for q in ("fate", "ix", "nScatter"))) for i in bad[:6]])
    integ.finalize()
