import sys, numpy as np
sys.path.insert(0, '/root/repo')
from tests import cases
from tests.test_analytic import isotropic_slab, slab, SCATTERING_SLABS, SEED, _sigma
import mcbrat3d_amd as M
from mcbrat3d_amd.integrator import new_RandomNumberSequence
for b, omega, mu0 in SCATTERING_SLABS:
    case = slab(b, omega, nz=16)
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=True)
    photons = M.new_PhotonStream(mu0, 75.0, numberOfPhotons=10 ** 12)
    integ.resetMoments()
    ppb, nb = 10 ** 7, 100
    n = integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, ppb, nb)
    from mcbrat3d_amd import driver
    st = driver.statistics(driver.unpack_moments(integ.moments(), dom.numX, dom.numY, dom.numZ))
    up, down, direct = isotropic_slab(b, omega, mu0, cells=3000)
    print("b %.1f omega %.1f mu0 %.1f n %.0e: up %.6f +- %.6f theory %.6f z %.2f | down %.6f +- %.6f theory %.6f z %.2f | bad %d" % (
        b, omega, mu0, n, st["meanFluxUp"], st["meanFluxUp_StdErr"], up, (st["meanFluxUp"] - up) / st["meanFluxUp_StdErr"],
        st["meanFluxDown"], st["meanFluxDown_StdErr"], down + direct, (st["meanFluxDown"] - down - direct) / st["meanFluxDown_StdErr"], integ.badPhotons()))
    integ.finalize()
