"""Development timing: radiance on the 128x128x64 cloud field (layer-skipping walk) over event / launch / exit thresholds.
usage: python scripts/inten_sweep.py [launches of 1e6 photons]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import cases
import mcbrat3d_amd as M
from mcbrat3d_amd.integrator import new_RandomNumberSequence
dom = cases.product_domain(cases.landsat_like())
photons = M.new_PhotonStream(0.5, 30.0, numberOfPhotons=10 ** 12)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for ndir in (1, 4):
    for thr, lthr, sthr in ((0, 0, 0), (16, 0, 0), (24, 0, 0), (32, 0, 0), (40, 0, 0), (48, 0, 0), (32, 16, 0), (32, 32, 0), (32, 0, 24), (32, 0, 4)):
        integ = M.new_Integrator(dom)
        mus = np.linspace(1.0, 0.3, ndir); phis = np.linspace(0.0, 300.0, ndir)
        integ.specifyParameters(minInverseTableSize=9001, intensityMus=mus, intensityPhis=phis, computeIntensity=True, useRussianRouletteForIntensity=True)
        integ.setTuning(eventThreshold=thr, launchThreshold=lthr, surfaceThreshold=sthr)
        rng = new_RandomNumberSequence(5)
        integ.resetMoments()
        integ.computeRadiativeTransfer(dom, rng, photons, 200000, 5)
        n = integ.computeRadiativeTransfer(dom, rng, photons, 1000000, NB)
        print("ndir=%d thr=%d (in use %d) lthr=%d sthr=%d: kernel %.1f ms  %.3g photons/s" % (ndir, thr, integ.eventThreshold(), lthr, sthr, integ.lastTraceMs(), n / (integ.lastTraceMs() * 1e-3)), flush=True)
        integ.finalize()
