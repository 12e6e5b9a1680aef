"""Is the max column |z| on the 128x128x64 domain an estimator artefact?  Compare GPU run A vs GPU run B
(different photons, same code) with the same batch statistics."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from tests import cases  # noqa: E402
import mcbrat3d_amd as M  # noqa: E402
from mcbrat3d_amd import driver  # noqa: E402
from mcbrat3d_amd.integrator import new_RandomNumberSequence  # noqa: E402

case = cases.landsat_like()
dom = cases.product_domain(case)
integ = M.new_Integrator(dom)
integ.specifyParameters(minInverseTableSize=10001)
photons = M.new_PhotonStream(0.5, 30.0, numberOfPhotons=10 ** 12)
res = []
for first, ppb, nb in ((0, 1000000, 20), (10 ** 11, 100000, 64)):
    integ.resetMoments()
    integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(10, first), photons, ppb, nb)
    res.append(driver.statistics(driver.unpack_moments(integ.moments(), 128, 128, 64)))
a, b = res
for k in ("fluxUp", "fluxDown", "fluxAbsorbed"):
    z = (a[k] - b[k]) / np.sqrt(a[k + "_StdErr"] ** 2 + b[k + "_StdErr"] ** 2 + 1e-30)
    print(k, "GPU(20x1e6) vs GPU(64x1e5): max|z| %.2f  frac>3 %.5f  frac>4 %.6f  mean %.4f std %.3f" % (
        np.abs(z).max(), (np.abs(z) > 3).mean(), (np.abs(z) > 4).mean(), z.mean(), z.std()))
