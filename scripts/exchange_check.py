"""Development check: the photon-exchange kernel against the one-photon-per-lane kernel (bitwise) and its rate."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from tests import cases  # noqa: E402

import mcbrat3d_amd as M  # noqa: E402
from mcbrat3d_amd.integrator import new_RandomNumberSequence  # noqa: E402


def run(case, mu0, phi0, ppb, nb, exchange, reps=1, tuning=None):
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=10001)
    integ.setTuning(eventThreshold=32, exchange=exchange, **(tuning or {}))
    photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 12)
    best = 0.0
    for r in range(reps):
        integ.resetMoments()
        n = integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(77), photons, ppb, nb)
        best = max(best, n / (integ.lastTraceMs() * 1e-3))
    mom = integ.moments().copy()
    integ.finalize()
    return mom, best


which = sys.argv[1] if len(sys.argv) > 1 else "small"
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # 1: photon-exchange kernel
if which == "small":
    todo = [("landsat48", cases.landsat_like(n=48, nz=24, n_entries=6), 0.5, 30.0, 20000, 3, None),
            ("landsat48 albedo", cases.landsat_like(n=48, nz=24, n_entries=6, albedo=0.4), 0.5, 30.0, 20000, 3, None),
            ("step priv2", cases.step_cloud(0.99), 1.0, 0.0, 20000, 3, None),
            ("step priv1", cases.step_cloud(0.99), 1.0, 0.0, 20000, 3, dict(privateTallies=2)),
            ("step global", cases.step_cloud(0.99), 0.5, 30.0, 20000, 3, dict(privateTallies=0)),
            ("stretched", cases.stretched_grid_cloud(), 0.6, 75.0, 20000, 3, None),
            ("ragged", cases.step_cloud(0.99), 1.0, 0.0, 777, 5, None)]
    reps = 1
elif which == "landsat":
    todo = [("landsat128", cases.landsat_like(), 0.5, 30.0, 1000000, 100, None)]
    reps = 3
else:
    todo = [("step", cases.step_cloud(0.99), 1.0, 0.0, 100000, 100, None)]
    reps = 3
for name, case, mu0, phi0, ppb, nb, tuning in todo:
    a, ra = run(case, mu0, phi0, ppb, nb, 0, reps, tuning)
    print("%-18s lane kernel     %.3g photons/s" % (name, ra), flush=True)
    b, rb = run(case, mu0, phi0, ppb, nb, mode, reps, tuning)
    print("%-18s mode %d kernel   %.3g photons/s   bitwise equal: %s  (max |diff| %.3g)" % (name, mode, rb, np.array_equal(a, b), np.max(np.abs(a - b))), flush=True)
