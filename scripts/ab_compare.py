"""A/B helper (development): moments of a workload under the library given by MCBRAT_LIB; prints a checksum + rate."""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases  # noqa: E402
import mcbrat3d_amd as M  # noqa: E402
from mcbrat3d_amd.integrator import new_RandomNumberSequence  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "landsat"
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 32
case = {"landsat": cases.landsat_like, "radar": cases.radar_like}[name]()
dom = cases.product_domain(case)
integ = M.new_Integrator(dom)
integ.specifyParameters(minInverseTableSize=10001)
integ.setTuning(eventThreshold=thr)
photons = M.new_PhotonStream(0.5, 30.0, numberOfPhotons=10 ** 15)
integ.resetMoments()
integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(7), photons, 1000000, 20)
h = hashlib.sha256(integ.moments().tobytes()).hexdigest()[:16]
best = 0.0
for r in range(3):
    integ.resetMoments()
    t = time.time()
    n = integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(100 + r), photons, 1000000, 100)
    best = max(best, n / (time.time() - t))
print("lib=%s case=%s thr=%d moments sha %s  rate %.3g ph/s  kernel %.2f ms" % (os.path.basename(os.environ.get("MCBRAT_LIB", "default")), name, thr, h, best, integ.lastTraceMs()))
