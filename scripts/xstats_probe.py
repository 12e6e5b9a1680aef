import os, sys
sys.path.insert(0, "/root/repo")
from tests import cases
import mcbrat3d_amd as M
from mcbrat3d_amd.integrator import new_RandomNumberSequence
which = sys.argv[1]
case = cases.landsat_like() if which == "landsat" else cases.step_cloud(0.99)
mu0, phi0, ppb, nb = (0.5, 30.0, 1000000, 100) if which == "landsat" else (1.0, 0.0, 100000, 100)
dom = cases.product_domain(case)
integ = M.new_Integrator(dom)
integ.specifyParameters(minInverseTableSize=10001)
integ.setTuning(eventThreshold=32, exchange=1)
photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 12)
integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(77), photons, ppb, nb)
integ.enableCounters(True)
integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(78), photons, ppb, nb)
print("kernel ms", integ.lastTraceMs())
