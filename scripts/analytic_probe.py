"""The product against transport theory at 10^9 photons per slab (tests/test_analytic.py holds the solvers and runs the same
comparison at 4x10^6): isotropic slabs vs the integral equation, Henyey-Greenstein slabs vs matrix doubling
(for the phase function the reference samples, staircase lookup included: tests/test_analytic.py sampled_moments)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases
from tests.test_analytic import (isotropic_slab, doubling_slab, sampled_moments, slab, hg_slab, tabulated_slab, SCATTERING_SLABS,
                                 HG_SLABS, HG_STREAMS, SEED)
import mcbrat3d_amd as M
from mcbrat3d_amd import driver
from mcbrat3d_amd.integrator import new_RandomNumberSequence


def run(case, mu0, phi0, label, up, down):
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=True)
    photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 12)
    integ.resetMoments()
    n = integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, 10 ** 7, int(os.environ.get("ANALYTIC_BATCHES", "100")))
    st = driver.statistics(driver.unpack_moments(integ.moments(), dom.numX, dom.numY, dom.numZ))
    print("%s n %.0e: up %.6f +- %.6f theory %.6f z %.2f | down %.6f +- %.6f theory %.6f z %.2f | bad %d" % (
        label, n, st["meanFluxUp"], st["meanFluxUp_StdErr"], up, (st["meanFluxUp"] - up) / st["meanFluxUp_StdErr"],
        st["meanFluxDown"], st["meanFluxDown_StdErr"], down, (st["meanFluxDown"] - down) / st["meanFluxDown_StdErr"],
        integ.badPhotons()), flush=True)
    integ.finalize()


for b, omega, mu0 in SCATTERING_SLABS:
    up, down, direct = isotropic_slab(b, omega, mu0, cells=3000)
    run(slab(b, omega, nz=16), mu0, 75.0, "isotropic b %.1f omega %.2f mu0 %.3f" % (b, omega, mu0), up, down + direct)
for b, omega, g, nleg, node in HG_SLABS:
    case, chi = hg_slab(b, omega, g, nleg)
    mu0, up, down = doubling_slab(b, omega, sampled_moments(chi, table=9001), node, streams=HG_STREAMS)
    run(case, mu0, 20.0, "HG g %.2f (%d terms) b %.1f omega %.2f mu0 %.4f" % (g, nleg, b, omega, mu0), up, down)
case, nodes = tabulated_slab(2.0, 0.9)
mu0, up, down = doubling_slab(2.0, 0.9, sampled_moments(None, table=9001, nodes=nodes), 64, streams=HG_STREAMS)
run(case, mu0, 20.0, "angle / value table (two lobes, 361 angles) b 2.0 omega 0.90 mu0 %.4f" % mu0, up, down)
