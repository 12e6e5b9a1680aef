"""Step cloud: how long one leg of a lone history takes (the serial chain that sets the drain of a launch).  A launch of N photons in ONE batch
(few workgroups busy, each lane at most a few photons): production kernel time against the longest history in it (legs from traceFates)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import cases
import mcbrat3d_amd as M
from mcbrat3d_amd.integrator import new_RandomNumberSequence
dom = cases.product_domain(cases.step_cloud(0.99))
integ = M.new_Integrator(dom)
integ.specifyParameters(minInverseTableSize=10001)
integ.setTuning(eventThreshold=16)
photons = M.new_PhotonStream(1.0, 0.0, numberOfPhotons=10 ** 15)
for n in (64, 768, 6144, 49152, 393216):
    for seed0 in (0, 10 ** 9, 2 * 10 ** 9):
        rng = new_RandomNumberSequence(10); rng.nextPhotonId = seed0
        f = integ.traceFates(dom, rng, photons, n)
        legs = f["nEvents"]
        ts = []
        for rep in range(4):
            rng = new_RandomNumberSequence(10); rng.nextPhotonId = seed0; photons.currentPhoton = 1
            integ.resetMoments(); integ.computeRadiativeTransfer(dom, rng, photons, n, 1); ts.append(integ.lastTraceMs())
        t = min(ts[1:])
        print("n %7d: kernel %.3f ms, legs mean %.1f max %d sum/768 lanes %.0f -> us per leg of the longest %.3f" % (n, t, legs.mean(), legs.max(), legs.sum() / 768.0, 1e3 * t / legs.max()), flush=True)
integ.finalize()
