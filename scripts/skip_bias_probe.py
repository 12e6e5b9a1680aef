"""Development probe: domain means of 1e9 photons on the same Philox streams with the face-by-face walk (layerSkip 0), with the
layer-skipping walk only (2) and with the clear-air flight as well (1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import cases
import mcbrat3d_amd as M
from mcbrat3d_amd import driver
from mcbrat3d_amd.integrator import new_RandomNumberSequence
case = cases.landsat_like()
dom = cases.product_domain(case)
out = {}
for skip in (0, 2, 1):
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=10001)
    integ.setTuning(eventThreshold=24, layerSkip=skip)
    photons = M.new_PhotonStream(0.5, 30.0, numberOfPhotons=10 ** 15)
    integ.resetMoments()
    rng = new_RandomNumberSequence(10, 10 ** 12)
    for i in range(10):
        integ.computeRadiativeTransfer(dom, rng, photons, 1000000, 100)
    st = driver.statistics(driver.unpack_moments(integ.moments(), dom.numX, dom.numY, dom.numZ))
    out[skip] = st
    print("skip=%d photons %d means %.8f %.8f %.8f  stderr %.2e %.2e %.2e" % (skip, st["totalPhotons"], st["meanFluxUp"], st["meanFluxDown"], st["meanFluxAbsorbed"],
          st["meanFluxUp_StdErr"], st["meanFluxDown_StdErr"], st["meanFluxAbsorbed_StdErr"]), flush=True)
    integ.finalize()
for a in (2, 1):
    print("difference skip%d - skip0: %.3e %.3e %.3e" % ((a,) + tuple(out[a][k] - out[0][k] for k in ("meanFluxUp", "meanFluxDown", "meanFluxAbsorbed"))))
prof = out[1]["absorptionProfile"] - out[0]["absorptionProfile"] if "absorptionProfile" in out[1] else None
if prof is not None:
    rel = prof / np.maximum(out[0]["absorptionProfile"], 1e-30)
    print("absorption profile relative difference, max |.| %.3e, layers with largest:" % np.max(np.abs(rel)), np.argsort(-np.abs(rel))[:5], rel[np.argsort(-np.abs(rel))[:5]])
