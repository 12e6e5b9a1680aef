"""Development timing of the radiance path: photons/s with intensity directions on the GPU and in the CPU oracle."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from tests import cases  # noqa: E402
import mcbrat3d_amd as M  # noqa: E402
from mcbrat3d_amd.integrator import new_RandomNumberSequence  # noqa: E402

for name, case, mu0, phi0, ppb, nb in (("step", cases.step_cloud(0.99), 1.0, 0.0, 100000, 20),
                                       ("landsat64", cases.landsat_like(n=64, nz=32), 0.5, 30.0, 200000, 10)):
    dom = cases.product_domain(case)
    photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 12)
    for ndir in (0, 1, 4, 16):
        for rr in ((False,) if ndir == 0 else (False, True)):
            integ = M.new_Integrator(dom)
            mus = np.linspace(1.0, 0.3, max(ndir, 1))[:ndir]
            phis = np.linspace(0.0, 300.0, max(ndir, 1))[:ndir]
            integ.specifyParameters(minInverseTableSize=9001, intensityMus=mus, intensityPhis=phis, computeIntensity=ndir > 0,
                                    useRussianRouletteForIntensity=rr)
            rng = new_RandomNumberSequence(5)
            integ.resetMoments()
            integ.computeRadiativeTransfer(dom, rng, photons, ppb, nb)  # warm-up + threshold choice
            t0 = time.time()
            n = integ.computeRadiativeTransfer(dom, rng, photons, ppb, nb)
            dt = time.time() - t0
            print("%s ndir=%2d roulette=%d: %.3g photons/s (kernel %.1f ms)" % (name, ndir, rr, n / dt, integ.lastTraceMs()), flush=True)
            integ.finalize()
    if "--cpu" in sys.argv:
        from oracle import oracle as O
        P = cases.oracle_problem(case, nsteps=9001)
        for ndir, rr in ((0, False), (4, False), (4, True)):
            mus = np.linspace(1.0, 0.3, max(ndir, 1))[:ndir]
            phis = np.linspace(0.0, 300.0, max(ndir, 1))[:ndir]
            n = 20000
            t0 = time.time()
            if ndir:
                I = cases.oracle_intensity(case, mus, phis, use_russian_roulette=rr)
                O.compute_rt_intensity(P, O.solar_source(mu0, phi0), O.mt_rng(3), n, I)
            else:
                O.compute_rt(P, O.solar_source(mu0, phi0), O.mt_rng(3), n)
            print("%s oracle 1 core ndir=%d roulette=%d: %.3g photons/s" % (name, ndir, rr, n / (time.time() - t0)), flush=True)
