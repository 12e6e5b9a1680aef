"""The photon-exchange form of the tracing kernel (mcbrat_set_walk_options(exchange = 1), mcbrat_exchange.hip)
through the C ABI on the GPU: same Philox slots, same arithmetic, integer tallies -- so the moment arrays must be
BITWISE those of the one-photon-per-lane kernel, for every tally mode, for photon counts far below one wave and
for ragged batches."""
import numpy as np
import pytest

from tests import cases

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300, method="thread")]


@pytest.fixture(scope="module")
def M():
    import mcbrat3d_amd
    return mcbrat3d_amd


def _moments(M, case, mu0, phi0, ppb, nb, exchange, **tuning):
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=True)
    integ.setTuning(eventThreshold=24, exchange=exchange, **tuning)
    photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 12)
    integ.resetMoments()
    n = integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(991), photons, ppb, nb)
    mom = integ.moments().copy()
    integ.finalize()
    assert n == ppb * nb
    return mom


@pytest.mark.parametrize("name,ppb,nb,tuning", [
    ("cloud field, global tallies", 20000, 3, {}),
    ("cloud field, reflecting surface", 20000, 2, {}),
    ("step cloud, grid and tallies in LDS", 20000, 3, {}),
    ("step cloud, private tallies only", 5000, 3, dict(privateTallies=2)),
    ("step cloud, global tallies", 5000, 3, dict(privateTallies=0)),
    ("stretched grid", 20000, 2, {}),
    ("one photon", 1, 1, {}),
    ("63 photons in 7 batches", 9, 7, {}),
    ("ragged batches", 777, 5, {}),
])
def test_exchange_kernel_is_bitwise_the_lane_kernel(M, name, ppb, nb, tuning):
    if name.startswith("cloud field"):
        case = cases.landsat_like(n=48, nz=24, n_entries=6, albedo=0.4 if "surface" in name else 0.0)
        mu0, phi0 = 0.5, 30.0
    elif name == "stretched grid":
        case, mu0, phi0 = cases.stretched_grid_cloud(), 0.6, 75.0
    else:
        case, mu0, phi0 = cases.step_cloud(0.99), 1.0, 0.0
    a = _moments(M, case, mu0, phi0, ppb, nb, 0, **tuning)
    b = _moments(M, case, mu0, phi0, ppb, nb, 1, **tuning)
    assert a[0] == ppb * nb and np.array_equal(a, b), name
