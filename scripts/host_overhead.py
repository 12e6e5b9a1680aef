"""Where a synchronous step's wall time goes on the host side (step cloud, 1e7 photons per step).
usage: python scripts/host_overhead.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases  # noqa: E402
import mcbrat3d_amd as M  # noqa: E402
from mcbrat3d_amd.integrator import new_RandomNumberSequence  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dom = cases.product_domain(cases.step_cloud(0.99))
for spin in (0, 5000):
    os.environ["MCBRAT_SPIN_US"] = str(spin)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=10001)
    integ.setTuning(eventThreshold=16)
    photons = M.new_PhotonStream(1.0, 0.0, numberOfPhotons=10 ** 15)
    rng = new_RandomNumberSequence(10)
    for i in range(3):
        integ.resetMoments(); integ.computeRadiativeTransfer(dom, rng, photons, 100000, 100)
    t_reset = t_comp = kms = 0.0
    t0 = time.perf_counter()
    for i in range(steps):
        photons.currentPhoton = 1
        a = time.perf_counter(); integ.resetMoments(); b = time.perf_counter()
        integ.computeRadiativeTransfer(dom, rng, photons, 100000, 100); c = time.perf_counter()
        t_reset += b - a; t_comp += c - b; kms += integ.lastTraceMs()
    wall = time.perf_counter() - t0
    print("spin %5d us: per step wall %.1f us = resetMoments %.1f + computeRadiativeTransfer %.1f (tracing kernel %.1f) -> %.4g photons/s" % (
        spin, 1e6 * wall / steps, 1e6 * t_reset / steps, 1e6 * t_comp / steps, 1e3 * kms / steps, 1e7 * steps / wall), flush=True)
    integ.finalize()
