"""Development timing: radiance on the 128x128x64 cloud field, with and without layer skipping."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import cases
import mcbrat3d_amd as M
from mcbrat3d_amd.integrator import new_RandomNumberSequence
dom = cases.product_domain(cases.landsat_like())
photons = M.new_PhotonStream(0.5, 30.0, numberOfPhotons=10 ** 12)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 20  # launches of NB x 1e6 photons: small launches measure the tail of the longest history, not the rate
for ndir, rr in ((0, False), (1, True), (4, True), (4, False)):
    for skip in (0, 1):
        integ = M.new_Integrator(dom)
        mus = np.linspace(1.0, 0.3, max(ndir, 1))[:ndir]; phis = np.linspace(0.0, 300.0, max(ndir, 1))[:ndir]
        integ.specifyParameters(minInverseTableSize=9001, intensityMus=mus, intensityPhis=phis, computeIntensity=ndir > 0,
                                useRussianRouletteForIntensity=rr)
        integ.setTuning(eventThreshold=32, layerSkip=skip)
        rng = new_RandomNumberSequence(5)
        integ.resetMoments()
        integ.computeRadiativeTransfer(dom, rng, photons, 200000, 5)
        n = integ.computeRadiativeTransfer(dom, rng, photons, 1000000, NB)
        r = integ.reportResults()
        print("ndir=%d roulette=%d skip=%d: kernel %.1f ms  %.3g photons/s  mean radiance %s" % (ndir, rr, skip, integ.lastTraceMs(), n / (integ.lastTraceMs() * 1e-3), np.round(r["meanIntensity"], 5) if "meanIntensity" in r else ""), flush=True)
        integ.finalize()
