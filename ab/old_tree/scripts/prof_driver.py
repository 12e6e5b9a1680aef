"""Minimal launch sequence for rocprofv3 passes: set up one workload and trace a few steps."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--case", default="step")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--thr", type=int, default=0)
ap.add_argument("--priv", type=int, default=-1)
ap.add_argument("--block", type=int, default=-1)
ap.add_argument("--brick", type=int, default=-1)
a = ap.parse_args()
import mcbrat3d_amd as M  # noqa: E402
from mcbrat3d_amd.integrator import new_RandomNumberSequence  # noqa: E402

if a.case == "step":
    case, mu0, phi0, ppb, nb = cases.step_cloud(0.99), 1.0, 0.0, 100000, 100
else:
    case, mu0, phi0, ppb, nb = cases.landsat_like(), 0.5, 30.0, 1000000, 10
dom = cases.product_domain(case)
integ = M.new_Integrator(dom)
integ.specifyParameters(minInverseTableSize=10001)
integ.setTuning(eventThreshold=a.thr, privateTallies=a.priv, blockSize=a.block, brickLayout=a.brick)
photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 12)
rng = new_RandomNumberSequence(10)
for i in range(a.steps):
    integ.computeRadiativeTransfer(dom, rng, photons, ppb, nb)
print("trace ms", integ.lastTraceMs())
