"""BASELINE.json configs[2] (128x128x64 cloud field, 1e8 photons) and configs[4] (radar-like 128x128x64, roulette
heavy) at FULL grid size on the GPU: properties that need no reference run -- energy closure, independence of how
the photons are cut into batches and into calls (ranks), run-to-run bitwise equality -- and, on the full grid,
per-photon fates and one batch's fluxes against the oracle.  Run on the MI355X box with `-m gpu`."""
import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu
SEED = 4242


@pytest.fixture(scope="module")
def M():
    import mcbrat3d_amd
    return mcbrat3d_amd


def _moments(M, dom, mu0, phi0, calls, seed=SEED, tuning=None, options=None):
    """calls: list of (firstPhotonId, ppb, nb) traced into one moment array."""
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=10001, useRayTracing=True, useRussianRoulette=True)
    if tuning:
        integ.setTuning(**tuning)
    if options:
        integ.setOption(**options)
    photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 15)
    integ.resetMoments()
    for first, ppb, nb in calls:
        integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(seed, first), photons, ppb, nb)
    mom = integ.moments()
    integ.finalize()
    return mom


@pytest.mark.parametrize("name,make", [("config 3: cloud field 128x128x64", "landsat_like"),
                                       ("config 5: radar-like 128x128x64", "radar_like")])
def test_full_size_properties(M, name, make):
    from mcbrat3d_amd import driver
    case = getattr(cases, make)()
    dom = cases.product_domain(case)
    nx, ny, nz = dom.numX, dom.numY, dom.numZ
    ppb, nb = 10 ** 6, 100  # 1e8 photons, about 0.13 s of GPU time
    a = _moments(M, dom, 0.5, 30.0, [(0, ppb, nb)])
    st = driver.statistics(driver.unpack_moments(a, nx, ny, nz))
    assert st["totalPhotons"] == 10 ** 8 and st["batches"] == nb
    # energy closure, SW albedo 0 (monteCarloRadiativeTransfer.f95:221-223); roulette keeps it only statistically exact
    closure = st["meanFluxUp"] + st["meanFluxDown"] + st["meanFluxAbsorbed"]
    assert abs(closure - 1.0) < 3.0 / np.sqrt(1e8), closure
    # reportResults :881-884, :966: domain means are the means of the columns, the profile sums to the absorbed flux
    for k in ("meanFluxUp", "meanFluxDown", "meanFluxAbsorbed"):
        assert abs(st[k.replace("meanF", "f")].mean() - st[k]) < 2e-6
    dz = np.diff(case["ze"])
    assert abs(np.sum(st["absorbedProfile"] * dz) * 1000.0 - st["meanFluxAbsorbed"]) < 2e-5
    assert np.all(st["fluxUp"] >= 0) and np.all(st["fluxDown"] >= 0) and np.all(st["absorbedVolume"] >= 0)
    # the same photons in two calls of 50 batches (what two ranks would trace): the per-batch values are bitwise the
    # same (integer tallies), the f64 moment sums differ only by the association of the two partial sums
    b = _moments(M, dom, 0.5, 30.0, [(0, ppb, nb // 2), (ppb * (nb // 2), ppb, nb // 2)])
    assert np.array_equal(a[:8], b[:8])
    assert np.allclose(a, b, rtol=1e-13, atol=1e-6)  # (S1 entries are ~1e8 x flux: 1e-6 absolute is 1e-14 relative)
    # run to run
    c = _moments(M, dom, 0.5, 30.0, [(0, ppb, nb)])
    assert np.array_equal(a, c)
    # the same 1e8 photons cut into 10 batches of 1e7: the means move only by float32 normalisation
    d = driver.statistics(driver.unpack_moments(_moments(M, dom, 0.5, 30.0, [(0, 10 ** 7, 10)]), nx, ny, nz))
    for k in ("meanFluxUp", "meanFluxDown", "meanFluxAbsorbed"):
        assert abs(d[k] - st[k]) < 2e-6, (k, d[k], st[k])
    # a different seed agrees within the Monte Carlo error
    e = driver.statistics(driver.unpack_moments(_moments(M, dom, 0.5, 30.0, [(0, ppb, nb)], seed=SEED + 1), nx, ny, nz))
    for k in ("meanFluxUp", "meanFluxDown", "meanFluxAbsorbed"):
        assert abs(e[k] - st[k]) < 5 * np.sqrt(e[k + "_StdErr"] ** 2 + st[k + "_StdErr"] ** 2), k


def test_radar_like_full_grid_fates_and_batch_against_the_oracle(M):
    """Config 5 on the full 128x128x64 grid: 40 000 photons, same Philox streams in kernel and oracle (face-by-face
    walk: layerSkip off, so that per-photon identity is meaningful), then one batch's normalised results."""
    from oracle import oracle as O
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    case = cases.radar_like()
    n = 40000
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=10001, useRayTracing=True, useRussianRoulette=True)
    integ.setTuning(layerSkip=0)
    photons = M.new_PhotonStream(0.5, 30.0, numberOfPhotons=10 ** 9)
    got = integ.traceFates(dom, new_RandomNumberSequence(SEED), photons, n)
    P = cases.oracle_problem(case)
    ref = O.compute_rt(P, O.solar_source(0.5, 30.0), O.philox_rng(SEED, 0), n, want_fates=True)
    rf = ref["fates"]
    same = (got["fate"] == rf["fate"]) & (got["ix"] == rf["ix"]) & (got["iy"] == rf["iy"]) & \
        (got["nScatter"] == rf["nScatter"]) & (np.abs(got["weight"] - rf["weight"]) <= 1e-6)
    assert same.mean() > 0.985, "only %.4f of photon histories identical" % same.mean()
    assert (got["fate"] == 2).mean() > 0.3  # roulette kills a large share of the photons here (omega0 = 0.9)
    integ.setTuning(layerSkip=1)
    integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, n)
    res = integ.reportResults()
    norm = O.normalize(P, n, ref)
    mu, md, ma, prof = O.report_means(P, norm)
    for g, r in ((res["meanFluxUp"], mu), (res["meanFluxDown"], md), (res["meanFluxAbsorbed"], ma)):
        assert abs(g - r) < 6e-3 * max(r, 0.05), (g, r)
    assert np.allclose(res["absorbedProfile"], prof, rtol=0.04, atol=3e-5 * np.max(prof) + 1e-9)
    integ.finalize()

