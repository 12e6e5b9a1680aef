import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from tests import cases
import mcbrat3d_amd as M
from mcbrat3d_amd.integrator import new_RandomNumberSequence
t0 = time.time()
case = cases.landsat_like(n=512, nz=128)
dom = cases.product_domain(case)
print("domain built %.1fs" % (time.time() - t0), flush=True)
integ = M.new_Integrator(dom)
integ.specifyParameters(minInverseTableSize=9001)
photons = M.new_PhotonStream(0.5, 30.0, numberOfPhotons=10 ** 12)
for brick in (0, 1):
    integ.setTuning(eventThreshold=32, brickLayout=brick)
    rng = new_RandomNumberSequence(3)
    integ.resetMoments()
    t = time.time()
    n = integ.computeRadiativeTransfer(dom, rng, photons, 2000000, 10)
    dt = time.time() - t
    r = integ.reportResults()
    print("brick=%d: %d photons %.3g photons/s kernel %.1f ms, means %.5f %.5f %.5f closure %.2e" % (brick, n, n / dt, integ.lastTraceMs(), r["meanFluxUp"], r["meanFluxDown"], r["meanFluxAbsorbed"], r["meanFluxUp"] + r["meanFluxDown"] * (1 - dom.surfaceAlbedo) + r["meanFluxAbsorbed"] - 1), flush=True)
