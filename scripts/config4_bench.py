"""Config 4: broadband LW, isothermal homogeneous 20x20x20, 16 wavelengths 8-12 um,
1e8 photons split over the wavelengths by emitted power.  mcbrat3d_amd.broadband.SpectralRun: every wavelength's
optical properties and emission CDF are uploaded ONCE (prepare_thermal, outside the timed loop); the timed region is
the photon split on the device + the loop over (wavelength, batch) units: kernels only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from tests import cases  # noqa: E402
import mcbrat3d_amd as M  # noqa: E402
from mcbrat3d_amd import broadband, driver  # noqa: E402
from mcbrat3d_amd.integrator import new_RandomNumberSequence  # noqa: E402

nlam = 16
total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
ppb, nb = total // 100, 100
lambdas = np.linspace(8.0, 12.0, nlam)
t0 = time.time()
doms = [cases.product_domain(cases.homog_lw(n=20, lam=float(l))) for l in lambdas]
run = broadband.SpectralRun(M, doms, minInverseTableSize=9001)
flux = run.prepare_thermal(300.0)
t1 = time.time()
for rep in range(3):
    run.resetMoments()
    t2 = time.time()
    counts = run.run(ppb, nb, new_RandomNumberSequence(5), seed=3)
    mom = run.moments()  # (synchronises)
    t3 = time.time()
    print("run %d: %d photons over %d wavelengths in %.4f s -> %.3g photons/s (set-up + uploads, once: %.2f s)" % (
        rep, counts.sum(), nlam, t3 - t2, counts.sum() / (t3 - t2), t1 - t0), flush=True)
st = driver.statistics(driver.unpack_moments(mom, 20, 20, 20), solarFlux=flux)
print("flux up/down/absorbed W/m2: %.4f +- %.4f  %.4f +- %.4f  %.4f +- %.4f" % (st["meanFluxUp"], st["meanFluxUp_StdErr"], st["meanFluxDown"], st["meanFluxDown_StdErr"], st["meanFluxAbsorbed"], st["meanFluxAbsorbed_StdErr"]))
run.finalize()
