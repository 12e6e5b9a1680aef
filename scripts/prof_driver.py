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

ap_opts = dict(o.split("=") for o in os.environ.get("PROF_OPTIONS", "").split() if "=" in o)  # e.g. PROF_OPTIONS="jumpThreshold=16"
lw = a.case == "lw"
if a.case == "step":
    case, mu0, phi0, ppb, nb = cases.step_cloud(0.99), 1.0, 0.0, 100000, 100
elif lw:  # one wavelength of config 4 (20 x 20 x 20, thermal source): its share of a 1e8-photon step, 6.25e6 photons as 100 batches
    case, mu0, phi0, ppb, nb = cases.homog_lw(n=20, lam=10.0), 1.0, 0.0, 62500, 100
else:
    case, mu0, phi0, ppb, nb = cases.landsat_like(), 0.5, 30.0, 1000000, 10
dom = cases.product_domain(case)
integ = M.new_Integrator(dom)
integ.specifyParameters(minInverseTableSize=9001 if lw else 10001, **(dict(LW_flag=1.0) if lw else {}))
integ.setTuning(eventThreshold=a.thr, privateTallies=a.priv, blockSize=a.block, brickLayout=a.brick)
if ap_opts:
    integ.setOption(**{k: int(v) for k, v in ap_opts.items()})
if lw:
    w = M.new_Weights(dom.numX, dom.numY, dom.numZ)
    M.emission_weighting(dom, w, case["sfc_temp"])
    photons = M.new_PhotonStream(theseWeights=w, numberOfPhotons=10 ** 12)
else:
    photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 12)
rng = new_RandomNumberSequence(10)
for i in range(a.steps):
    integ.computeRadiativeTransfer(dom, rng, photons, ppb, nb)
print("trace ms", integ.lastTraceMs())
