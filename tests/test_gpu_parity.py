"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs.  Run on the MI355X box with `-m gpu`."""
import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu

SEED = 20240917


@pytest.fixture(scope="module")
def M():
    import mcbrat3d_amd
    return mcbrat3d_amd


def _setup(M, case, mu0=1.0, phi0=0.0, nsteps=10001, rr=True):
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=nsteps, useRayTracing=True, useRussianRoulette=rr)
    photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 9)
    return dom, integ, photons, new_RandomNumberSequence(SEED)


@pytest.mark.parametrize("ssa,mu0,phi0", [(0.99, 1.0, 0.0), (1.0, 0.5, 30.0)])
def test_step_cloud_fates_match_oracle(M, ssa, mu0, phi0):
    """Per-photon: same Philox streams -> same history, except where an ulp-level difference
    (device logf/cosf, parametric vs position-stepping walk) flips a discrete branch."""
    from oracle import oracle as O
    n = 50000
    case = cases.step_cloud(ssa=ssa)
    dom, integ, photons, rng = _setup(M, case, mu0, phi0)
    got = integ.traceFates(dom, rng, photons, n)
    P = cases.oracle_problem(case)
    ref = O.compute_rt(P, O.solar_source(mu0, phi0), O.philox_rng(SEED, 0), n, want_fates=True)["fates"]
    same = (got["fate"] == ref["fate"]) & (got["ix"] == ref["ix"]) & (got["iy"] == ref["iy"]) & \
        (got["nScatter"] == ref["nScatter"]) & (np.abs(got["weight"] - ref["weight"]) <= 1e-6)
    frac = same.mean()
    assert frac > 0.995, "only %.4f of photon histories identical" % frac
    # event counters of the two implementations agree to the same degree
    cg, cr = integ.counters(), O.compute_rt(P, O.solar_source(mu0, phi0), O.philox_rng(SEED, 0), n)["counters"]
    for k in ("legs", "collisions", "topExits", "surfaceHits"):
        assert abs(cg[k] - cr[k]) <= 0.002 * max(cr[k], 1) + 5, (k, cg[k], cr[k])


def test_step_cloud_batch_matches_oracle(M):
    """One computeRadiativeTransfer + reportResults against the oracle's, same photons."""
    from oracle import oracle as O
    n = 100000
    case = cases.step_cloud(ssa=0.99)
    dom, integ, photons, rng = _setup(M, case)
    done = integ.computeRadiativeTransfer(dom, rng, photons, n)
    assert done == n and rng.nextPhotonId == n
    got = integ.reportResults()
    P = cases.oracle_problem(case)
    ref = O.compute_radiative_transfer(P, O.solar_source(1.0, 0.0), O.philox_rng(SEED, 0), n)
    # a flipped photon moves 1/n * ncol in one column; allow a handful
    tol_col = 8.0 * 32 / n
    assert np.max(np.abs(got["fluxUp"][:, 0] - ref["fluxUp"])) < tol_col
    assert np.max(np.abs(got["fluxDown"][:, 0] - ref["fluxDown"])) < tol_col
    assert np.max(np.abs(got["fluxAbsorbed"][:, 0] - ref["fluxAbsorbed"])) < tol_col
    for k in ("meanFluxUp", "meanFluxDown", "meanFluxAbsorbed"):
        assert abs(got[k] - ref[k]) < 8.0 / n + 2e-6, (k, got[k], ref[k])
    vol_ref = ref["volumeAbsorption"].reshape(32, 1, 32).transpose(2, 1, 0)
    scale = np.max(np.abs(vol_ref))
    assert np.max(np.abs(got["volumeAbsorption"] - vol_ref)) < 0.02 * scale
    assert np.allclose(got["absorbedProfile"], ref["absorbedProfile"], rtol=5e-3, atol=1e-6)
    # energy closure, SW albedo 0 (monteCarloRadiativeTransfer.f95:221-223)
    assert abs(got["meanFluxUp"] + got["meanFluxDown"] + got["meanFluxAbsorbed"] - 1.0) < 3.0 / np.sqrt(n)


@pytest.fixture(scope="module")
def landsat():
    return cases.landsat_like()


def test_landsat_like_fates_and_batch(M, landsat):
    """128x128x64, irregular x/y/z grid, two components (multi-entry HG cloud table that does
    not fit LDS + 1-D Rayleigh), mu0 = 0.5, phi0 = 30 deg: per-photon and per-batch parity."""
    from oracle import oracle as O
    n = 40000
    dom, integ, photons, rng = _setup(M, landsat, 0.5, 30.0)
    got = integ.traceFates(dom, rng, photons, n)
    P = cases.oracle_problem(landsat)
    assert P.grid_flags()[:2] == (False, False)
    ref = O.compute_rt(P, O.solar_source(0.5, 30.0), O.philox_rng(SEED, 0), n, want_fates=True)
    rf = ref["fates"]
    same = (got["fate"] == rf["fate"]) & (got["ix"] == rf["ix"]) & (got["iy"] == rf["iy"]) & \
        (got["nScatter"] == rf["nScatter"]) & (np.abs(got["weight"] - rf["weight"]) <= 1e-6)
    assert same.mean() > 0.99, "only %.4f of photon histories identical" % same.mean()
    done = integ.computeRadiativeTransfer(dom, rng, photons, n)
    res = integ.reportResults()
    norm = O.normalize(P, n, ref)
    mu, md, ma, prof = O.report_means(P, norm)
    assert done == n
    for g, r in ((res["meanFluxUp"], mu), (res["meanFluxDown"], md), (res["meanFluxAbsorbed"], ma)):
        assert abs(g - r) < 5e-3 * max(r, 0.05), (g, r)
    assert np.allclose(res["absorbedProfile"], prof, rtol=0.03, atol=2e-5 * np.max(prof) + 1e-9)
    # column fluxes: each column sees ~2.4 photons, so compare the exact integer-like sums
    up_ref = norm["fluxUp"].reshape(128, 128).T
    assert abs(res["fluxUp"].sum() - up_ref.sum()) < 5e-3 * up_ref.sum()


def test_thermal_emission_matches_oracle(M):
    """LW path: emission_weighting -> BBEmission photon stream -> computeRT with LW_flag > 0
    (launch from the running voxel CDF or the surface, emission tallied as negative
    absorption, Lambertian surface albedo 0.1)."""
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    from oracle import oracle as O
    n = 60000
    case = cases.homog_lw(n=20)
    case["temps"] = case["temps"] + np.linspace(-15, 15, 20)[None, None, :]
    dom = cases.product_domain(case)
    w = M.new_Weights(20, 20, 20)
    M.emission_weighting(dom, w, case["sfc_temp"])
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=True, LW_flag=1.0)
    photons = M.new_PhotonStream(theseWeights=w, numberOfPhotons=10 ** 9)
    got = integ.traceFates(dom, new_RandomNumberSequence(SEED), photons, n)

    P = cases.oracle_problem(case, nsteps=9001, lw_flag=1.0)
    vw, frac, _ = O.emission_weighting(P, case["temps"].transpose(2, 1, 0).reshape(-1), case["lambda_um"], case["sfc_temp"])
    assert np.array_equal(vw, w.voxelWeights) and frac == w.fracAtmsPower
    src = O.EmissionSource(vw, frac)
    ref = O.compute_rt(P, src, O.philox_rng(SEED, 0), n, want_fates=True)
    rf = ref["fates"]
    same = (got["fate"] == rf["fate"]) & (got["ix"] == rf["ix"]) & (got["iy"] == rf["iy"]) & \
        (got["nScatter"] == rf["nScatter"]) & (np.abs(got["weight"] - rf["weight"]) <= 1e-6)
    assert same.mean() > 0.995, "only %.4f of photon histories identical" % same.mean()

    rng = new_RandomNumberSequence(SEED)
    assert integ.computeRadiativeTransfer(dom, rng, photons, n) == n
    res = integ.reportResults()
    norm = O.normalize(P, n, ref)
    mu, md, ma, prof = O.report_means(P, norm)
    for g, r in ((res["meanFluxUp"], mu), (res["meanFluxDown"], md), (res["meanFluxAbsorbed"], ma)):
        assert abs(g - r) < 4e-3 * max(abs(r), 0.05), (g, r)
    assert ma < 0  # the layer emits more than it absorbs
    assert np.allclose(res["absorbedProfile"], prof, rtol=0.05, atol=0.02 * np.max(np.abs(prof)))


def test_batch_moments_and_split_independence(M):
    """Moments over several batches equal the oracle's per-batch results folded with the
    driver's formulas; the same photons traced in one call or split over two calls (what
    two GPUs would each do) give bitwise identical moment arrays; reruns are bitwise equal."""
    from mcbrat3d_amd import driver
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    from oracle import oracle as O
    ppb, nb = 20000, 4
    case = cases.step_cloud(ssa=0.99)
    dom, integ, photons, _ = _setup(M, case)

    integ.resetMoments()
    rng = new_RandomNumberSequence(SEED)
    integ.computeRadiativeTransfer(dom, rng, photons, ppb, nb)
    whole = integ.moments()
    integ.resetMoments()
    rng = new_RandomNumberSequence(SEED)
    integ.computeRadiativeTransfer(dom, rng, photons, ppb, nb)
    assert np.array_equal(whole, integ.moments())  # fixed-point tallies: run-to-run reproducible

    parts = []
    for lo in (0, 2):  # "rank 0" takes batches 0-1, "rank 1" batches 2-3
        integ.resetMoments()
        rng = new_RandomNumberSequence(SEED, firstPhotonId=lo * ppb)
        integ.computeRadiativeTransfer(dom, rng, photons, ppb, 2)
        parts.append(integ.moments())
    assert np.allclose(parts[0] + parts[1], whole, rtol=1e-14, atol=0)

    P = cases.oracle_problem(case)
    batches_mean, batches_col = [], []
    for b in range(nb):
        r = O.compute_radiative_transfer(P, O.solar_source(1.0, 0.0), O.philox_rng(SEED, b * ppb), ppb)
        batches_mean.append((ppb, np.array([r["meanFluxUp"], r["meanFluxDown"], r["meanFluxAbsorbed"]], np.float64)))
        batches_col.append((ppb, r["fluxDown"].astype(np.float64)))
    mean, err = O.batch_statistics(batches_mean)
    cmean, cerr = O.batch_statistics(batches_col)
    st = driver.statistics(driver.unpack_moments(whole, 32, 1, 32))
    assert st["totalPhotons"] == ppb * nb and st["batches"] == nb
    got = np.array([st["meanFluxUp"], st["meanFluxDown"], st["meanFluxAbsorbed"]])
    assert np.all(np.abs(got - mean) < 5.0 / (ppb * nb) + 2e-6)
    goterr = np.array([st["meanFluxUp_StdErr"], st["meanFluxDown_StdErr"], st["meanFluxAbsorbed_StdErr"]])
    assert np.allclose(goterr, err, rtol=0.1, atol=2e-4)
    assert np.max(np.abs(st["fluxDown"][:, 0] - cmean)) < 8.0 * 32 / (ppb * nb)


def test_plane_parallel_plumbing(M):
    """Config 1: 1x1x32 plane-parallel slab, conservative and absorbing, two sun angles: every
    photon lands in the single column (worst case for tally contention).  With one column the
    reference's float32 tallies lose precision (1e5 weights of ~0.98 added into a float that has
    grown past 6.5e4 -- SURVEY.md 8a quirk 6), so the GPU's exact fixed-point sums are compared
    with the double-precision sum of the oracle's per-photon records; the float32-accumulated
    oracle result is only required to be within its own rounding error of that."""
    from oracle import oracle as O
    n = 100000
    for ssa, mu0 in ((1.0, 1.0), (0.99, 0.5)):
        case = cases.plane_parallel(ssa=ssa)
        dom, integ, photons, rng = _setup(M, case, mu0, 0.0)
        integ.computeRadiativeTransfer(dom, rng, photons, n)
        res = integ.reportResults()
        P = cases.oracle_problem(case)
        raw = O.compute_rt(P, O.solar_source(mu0, 0.0), O.philox_rng(SEED, 0), n, want_fates=True)
        f = raw["fates"]
        up = f["weight"][f["fate"] == 0].astype(np.float64).sum() / n
        down = f["weight"][f["fate"] == 1].astype(np.float64).sum() / n
        assert abs(res["meanFluxUp"] - up) < 5.0 / n + 2e-6, (ssa, mu0, res["meanFluxUp"], up)
        assert abs(res["meanFluxDown"] - down) < 5.0 / n + 2e-6, (ssa, mu0, res["meanFluxDown"], down)
        assert abs(res["meanFluxAbsorbed"] - (1.0 - up - down)) < 5.0 / n + 2e-6
        ref = O.normalize(P, n, raw)
        assert abs(ref["fluxDown"][0] - down) < 5e-4  # the reference's float32 accumulation error
        if ssa == 1.0:
            assert res["meanFluxAbsorbed"] == 0.0 and abs(res["meanFluxUp"] + res["meanFluxDown"] - 1.0) < 1e-6
        integ.finalize()


def test_radar_like_roulette_heavy(M):
    """Config 5 (reduced to 64x64x32 for the oracle's sake): tau up to ~100, omega0 = 0.9 -- most photons
    end at Russian roulette.  Per-photon and per-batch parity, and roulette's weight bookkeeping:
    kills and survivals balance statistically (:805-811)."""
    from oracle import oracle as O
    n = 40000
    case = cases.radar_like(n=64, nz=32)
    dom, integ, photons, rng = _setup(M, case, 0.7, 200.0)
    got = integ.traceFates(dom, rng, photons, n)
    cnt = integ.counters()
    P = cases.oracle_problem(case)
    ref = O.compute_rt(P, O.solar_source(0.7, 200.0), O.philox_rng(SEED, 0), n, want_fates=True)
    rf = ref["fates"]
    same = (got["fate"] == rf["fate"]) & (got["ix"] == rf["ix"]) & (got["iy"] == rf["iy"]) & \
        (got["nScatter"] == rf["nScatter"]) & (np.abs(got["weight"] - rf["weight"]) <= 1e-6)
    assert same.mean() > 0.99, "only %.4f of photon histories identical" % same.mean()
    assert (rf["fate"] == 2).mean() > 0.15  # roulette really is the common ending here
    for k in ("rouletteKills", "rouletteSurvivals", "collisions", "legs"):
        assert abs(cnt[k] - ref["counters"][k]) <= 0.01 * ref["counters"][k] + 5, (k, cnt[k], ref["counters"][k])
    integ.computeRadiativeTransfer(dom, rng, photons, n)
    res = integ.reportResults()
    mu, md, ma, _ = O.report_means(P, O.normalize(P, n, ref))
    for g, r in ((res["meanFluxUp"], mu), (res["meanFluxDown"], md), (res["meanFluxAbsorbed"], ma)):
        assert abs(g - r) < 6e-3 * max(r, 0.05), (g, r)
    assert abs(res["meanFluxUp"] + res["meanFluxDown"] + res["meanFluxAbsorbed"] - 1.0) < 0.02


def test_broadband_thermal_loop(M):
    """Config 4's mechanism: a loop over wavelength domains (8-12 um, isothermal 20x20x20 layer over
    a warmer surface), photons split by emitted power, one moment array for the whole spectrum.
    Oracle side: the same loop with the oracle's emission weighting and photon loop, same photon
    ids.  Parity unpinned for the driver-level pieces (restated from source text)."""
    from mcbrat3d_amd import broadband, driver
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    from oracle import oracle as O
    lambdas = [8.0, 9.0, 10.0, 11.0, 12.0]
    ppb, nb = 4000, 10
    doms, cases_l = [], []
    for lam in lambdas:
        c = cases.homog_lw(n=20, lam=lam, ext=5.0 + 0.5 * (lam - 8.0), ssa=0.5)
        cases_l.append(c)
        doms.append(cases.product_domain(c))
    integ = M.new_Integrator(doms[0])
    integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=True, LW_flag=1.0)
    integ.resetMoments()
    rng = new_RandomNumberSequence(SEED)
    counts, flux = broadband.run_thermal(integ, doms, 300.0, ppb, nb, rng, seed=3)
    assert counts.sum() == ppb * nb and rng.nextPhotonId == ppb * nb
    st = driver.statistics(driver.unpack_moments(integ.moments(), 20, 20, 20), solarFlux=flux)

    # oracle: same widths, fluxes, CDF; same photons per wavelength
    widths = broadband.spectral_widths(lambdas)
    assert np.allclose(widths, [1.0, 1.0, 1.0, 1.0, 1.0])
    fl, srcs, probs = [], [], []
    for c, dl in zip(cases_l, widths):
        P = cases.oracle_problem(c, nsteps=9001, lw_flag=1.0)
        vw, frac, f = O.emission_weighting(P, c["temps"].transpose(2, 1, 0).reshape(-1), c["lambda_um"], 300.0, dl)
        fl.append(f); srcs.append(O.EmissionSource(vw, frac)); probs.append(P)
    cdf, total = broadband.emitted_flux_cdf(fl)
    assert total == pytest.approx(flux, rel=1e-12) and cdf[-1] == 1.0 and np.all(np.diff(cdf) > 0)
    # photon split follows the power CDF (multinomial): 5-sigma check per wavelength
    p = np.diff(np.concatenate([[0.0], cdf]))
    assert np.all(np.abs(counts - p * ppb * nb) < 5 * np.sqrt(p * (1 - p) * ppb * nb) + 1)
    batches, first = [], 0
    for P, src, n in zip(probs, srcs, counts):
        left = int(n)
        while left > 0:
            k = min(ppb, left)
            r = O.compute_radiative_transfer(P, src, O.philox_rng(SEED, first), k)
            batches.append((k, np.array([r["meanFluxUp"], r["meanFluxDown"], r["meanFluxAbsorbed"]], np.float64)))
            first += k; left -= k
    mean, err = O.batch_statistics(batches, solar_flux=total)
    got = np.array([st["meanFluxUp"], st["meanFluxDown"], st["meanFluxAbsorbed"]])
    assert np.all(np.abs(got - mean) < 2e-3 * total), (got, mean)
    assert st["batches"] == len(batches)
    assert st["meanFluxAbsorbed"] < 0 < st["meanFluxUp"]  # the layer cools: it emits more than it absorbs


def test_stretched_grid_tabulated_phase_functions(M):
    """Non-uniform x/y/z grid (binary-search launch, per-cell edge table in the walk), reflecting
    surface, angle/value ("Mie-table" storage) phase functions next to a Legendre component, and an
    extinction field that changes from cell to cell.

    In such a medium photon histories are chaotic: the reference accumulates optical depth in
    float32 (opticalProperties.f95:1683, :1743), so two correct implementations stop a leg a few
    1e-7 km apart, and each later leg amplifies the offset whenever it moves the ray across a
    boundary between cells of different extinction (traced photon by photon with
    MCBRAT_TRACE_PHOTON / ORC_TRACE_PHOTON: 2e-7 km after one leg, 3e-4 km after six, another cell
    after eight).  Identical histories are therefore only required for the first leg (exact: same
    cell for every photon) and for most -- not all -- complete histories; the fluxes must agree
    statistically."""
    from oracle import oracle as O
    n = 40000
    case = cases.stretched_grid_cloud()
    # (1) first leg only: absorb everything at the first collision
    one = cases.stretched_grid_cloud()
    one["components"][0]["ssa"][:] = 0.0
    one["components"][1]["ssa"][:] = 0.0
    dom, integ, photons, rng = _setup(M, one, 0.6, 75.0)
    got = integ.traceFates(dom, rng, photons, n)
    P1 = cases.oracle_problem(one)
    rf = O.compute_rt(P1, O.solar_source(0.6, 75.0), O.philox_rng(SEED, 0), n, want_fates=True)["fates"]
    assert np.array_equal(got["fate"], rf["fate"]) and np.array_equal(got["ix"], rf["ix"])
    assert np.array_equal(got["iy"], rf["iy"]) and np.array_equal(got["iz"], rf["iz"])
    integ.finalize()
    # (2) complete histories
    dom, integ, photons, rng = _setup(M, case, 0.6, 75.0)
    got = integ.traceFates(dom, rng, photons, n)
    P = cases.oracle_problem(case)
    assert P.grid_flags()[:2] == (False, False)
    ref = O.compute_rt(P, O.solar_source(0.6, 75.0), O.philox_rng(SEED, 0), n, want_fates=True)
    rf = ref["fates"]
    same = (got["fate"] == rf["fate"]) & (got["ix"] == rf["ix"]) & (got["iy"] == rf["iy"]) & \
        (got["nScatter"] == rf["nScatter"]) & (np.abs(got["weight"] - rf["weight"]) <= 1e-6)
    assert same.mean() > 0.88, "only %.4f of photon histories identical" % same.mean()
    short = rf["nScatter"] <= 3
    assert same[short].mean() > 0.995  # short histories have no room to drift apart
    # (3) statistics: 4 batches of n photons each side, same photons
    integ.resetMoments()
    integ.computeRadiativeTransfer(dom, rng, photons, n, 4)
    from mcbrat3d_amd import driver
    st = driver.statistics(driver.unpack_moments(integ.moments(), 24, 10, 18))
    batches = []
    for b in range(4):
        r = O.compute_radiative_transfer(P, O.solar_source(0.6, 75.0), O.philox_rng(SEED, b * n), n)
        batches.append((n, np.array([r["meanFluxUp"], r["meanFluxDown"], r["meanFluxAbsorbed"]], np.float64)))
        if b == 3:
            last_ref = r
    mean, err = O.batch_statistics(batches)
    got3 = np.array([st["meanFluxUp"], st["meanFluxDown"], st["meanFluxAbsorbed"]])
    assert np.all(np.abs(got3 - mean) < 1.5e-3), (got3, mean)  # same photons: far inside the MC error (~2e-3)
    res = integ.reportResults()
    up_ref = last_ref["fluxUp"].reshape(10, 24).T  # irregular grid: columns normalised by their own area (:334-342)
    assert np.allclose(res["fluxUp"], up_ref, rtol=0.15, atol=0.06 * up_ref.max())
    assert np.allclose(res["absorbedProfile"], last_ref["absorbedProfile"], rtol=0.05, atol=5e-3 * np.max(last_ref["absorbedProfile"]))


def test_regular_grid_path(M):
    """Cell sizes exactly representable in float32 take the reference's 'regularly spaced' launch
    arithmetic (new_Integrator :163-181, findXYIndicies :1558-1569)."""
    from oracle import oracle as O
    n = 30000
    case = cases.landsat_like(n=32, nz=16, n_entries=4, regular=True)
    dom, integ, photons, rng = _setup(M, case, 0.8, 10.0)
    P = cases.oracle_problem(case)
    assert P.grid_flags()[:2] == (True, True)
    got = integ.traceFates(dom, rng, photons, n)
    rf = O.compute_rt(P, O.solar_source(0.8, 10.0), O.philox_rng(SEED, 0), n, want_fates=True)["fates"]
    same = (got["fate"] == rf["fate"]) & (got["ix"] == rf["ix"]) & (got["iy"] == rf["iy"]) & (got["nScatter"] == rf["nScatter"])
    assert same.mean() > 0.995


FUZZ = int(__import__("os").environ.get("MCBRAT_FLIGHT_FUZZ", "12"))  # seeds of the random differential test (raise it for a soak run)


# seeds the soak runs of this test found bugs with, kept whatever FUZZ is: 71 (a NaN drawn from the inverse table, DESIGN.md
# section 8, moved the block walk's photon along it for ever), 763 (a grazing sun: a wrap and a block face at the same step,
# the origin folded back across the domain boundary, the block walk went round a corner for ever)
SOAK_FINDS = (71, 763)


def check_event_totals(walk, case, cnt, got, rc, rf, same):
    """Event totals of product (counters `cnt`, per-photon records `got`) against oracle (`rc`, `rf`).  The histories that
    are identical to their end contribute identical counts by definition; the totals can only differ through the flipped
    ones, which -- long histories are chaotic, no roulette under a bright surface makes them long -- are in part independent
    samples of a heavy-tailed count (3 of 500 soak seeds differed by 2-3 % in the totals).  So: (i) each side's counters are
    exactly what its own per-photon records add up to, and (ii) over the flipped histories the differences must look like
    noise, not like a bias: the sum of the per-photon differences d_i against sqrt(sum d_i^2), 4 sigma -- no percentage to
    widen.  (Round 3 had this for the solar domains only; the thermal and the rich ones still carried a 5 % bound.)"""
    assert cnt["collisions"] + cnt["surfaceHits"] == int(got["nScatter"].sum()) and cnt["topExits"] == int((got["fate"] == 0).sum())
    assert cnt["legs"] == int(got["nEvents"].sum())
    assert rc["collisions"] + rc["surfaceHits"] == int(rf["nScatter"].sum()) and rc["topExits"] == int((rf["fate"] == 0).sum())
    for name, a, b in (("scatterings", got["nScatter"], rf["nScatter"]), ("top exits", got["fate"] == 0, rf["fate"] == 0),
                       ("surface absorptions", got["fate"] == 1, rf["fate"] == 1), ("roulette kills", got["fate"] == 2, rf["fate"] == 2)):
        d = a[~same].astype(np.float64) - b[~same].astype(np.float64)
        assert abs(d.sum()) <= 4.0 * np.sqrt((d * d).sum()) + 3.0, (walk, name, case["name"], d.sum(), np.sqrt((d * d).sum()), int((~same).sum()))
        # (and nothing outside the flipped histories: identical records have identical counts)
        assert np.array_equal(a[same], b[same]), (walk, name)


@pytest.mark.timeout(120, method="thread")
@pytest.mark.parametrize("seed", sorted(set(range(FUZZ)) | set(SOAK_FINDS)))
def test_random_domains_against_the_oracle(M, seed):
    """Differential test against the ORACLE (the restated reference walk, Philox mode) on random small domains: random
    grid (equal or stretched spacing, non-zero origin), random extinction with vacuum cells, one or two components, random
    sun, surface and roulette setting; every walk of the product in turn -- face by face, with the layer-skipping walk and
    the clear-air flight, and the block walk where the grid lives in LDS.  Histories are compared up to ten scatterings
    (longer ones are chaotic), event counters and fluxes over all of them."""
    from oracle import oracle as O
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    case, mu0, phi0, rr = random_oracle_case(seed)
    nx, ny, nz = len(case["xe"]) - 1, len(case["ye"]) - 1, len(case["ze"]) - 1
    n = 15000
    P = cases.oracle_problem(case, nsteps=9001, use_russian_roulette=rr)
    ref = O.compute_rt(P, O.solar_source(mu0, phi0), O.philox_rng(SEED, 0), n, want_fates=True)
    rf, order = ref["fates"], ref["fates"]["nScatter"]
    norm = O.normalize(P, n, ref)
    mu, md, ma, _ = O.report_means(P, norm)
    for walk, tuning in (("face by face", dict(privateTallies=0, layerSkip=0)), ("layers + flight", dict(privateTallies=0, layerSkip=3)),
                         ("LDS face by face", dict(blockWalk=0)), ("block walk", dict(blockWalk=2))):
        dom = cases.product_domain(case)
        integ = M.new_Integrator(dom)
        integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=rr)
        integ.setTuning(eventThreshold=16, **tuning)
        photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 9)
        got = integ.traceFates(dom, new_RandomNumberSequence(SEED), photons, n)
        cnt = integ.counters()
        integ.resetMoments()
        integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, n)
        res = integ.reportResults()
        integ.finalize()
        same = (got["fate"] == rf["fate"]) & (got["ix"] == rf["ix"]) & (got["iy"] == rf["iy"]) & (got["iz"] == rf["iz"]) & \
            (got["nScatter"] == rf["nScatter"]) & (np.abs(got["weight"] - rf["weight"]) <= 1e-6)
        assert same[order <= 10].mean() > 0.985, (walk, case["name"], nx, ny, nz, same[order <= 10].mean())
        check_event_totals(walk, case, cnt, got, ref["counters"], rf, same)
        for g, r in ((res["meanFluxUp"], mu), (res["meanFluxDown"], md), (res["meanFluxAbsorbed"], ma)):
            assert abs(g - r) < 5e-3, (walk, g, r)


@pytest.mark.timeout(120, method="thread")
@pytest.mark.parametrize("seed", range(FUZZ))
def test_random_thermal_domains_against_the_oracle(M, seed):
    """The same random domains as thermal sources (LW_flag > 0): random temperatures, surface temperature and wavelength.
    The emission weighting (the running voxel CDF and the atmosphere's share of the power) must equal the oracle's bit
    for bit; then every walk of the product against the oracle's photon loop launched from the same CDF: histories up to
    ten scatterings, event totals, mean fluxes (emission tallied as negative absorption)."""
    from oracle import oracle as O
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    case, _, _, rr = random_oracle_case(seed)
    rng = np.random.default_rng(52000 + seed)
    nx, ny, nz = len(case["xe"]) - 1, len(case["ye"]) - 1, len(case["ze"]) - 1
    case["temps"] = rng.uniform(230.0, 300.0, (nx, ny, nz))
    case["sfc_temp"] = float(rng.uniform(250.0, 320.0))
    case["lambda_um"] = float(rng.uniform(6.0, 14.0))
    n = 15000
    P = cases.oracle_problem(case, nsteps=9001, use_russian_roulette=rr, lw_flag=1.0)
    vw, frac, _ = O.emission_weighting(P, case["temps"].transpose(2, 1, 0).reshape(-1), case["lambda_um"], case["sfc_temp"])
    ref = O.compute_rt(P, O.EmissionSource(vw, frac), O.philox_rng(SEED, 0), n, want_fates=True)
    rf, order = ref["fates"], ref["fates"]["nScatter"]
    mu, md, ma, _ = O.report_means(P, O.normalize(P, n, ref))
    scale = max(abs(mu), abs(md), abs(ma), 1e-6)
    for walk, tuning in (("face by face", dict(privateTallies=0, layerSkip=0)), ("layers + flight", dict(privateTallies=0, layerSkip=3)),
                         ("LDS face by face", dict(blockWalk=0)), ("block walk", dict(blockWalk=2))):
        dom = cases.product_domain(case)
        w = M.new_Weights(nx, ny, nz)
        M.emission_weighting(dom, w, case["sfc_temp"])
        assert np.array_equal(vw, w.voxelWeights) and frac == w.fracAtmsPower
        integ = M.new_Integrator(dom)
        integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=rr, LW_flag=1.0)
        integ.setTuning(eventThreshold=16, **tuning)
        photons = M.new_PhotonStream(theseWeights=w, numberOfPhotons=10 ** 9)
        got = integ.traceFates(dom, new_RandomNumberSequence(SEED), photons, n)
        cnt = integ.counters()
        integ.resetMoments()
        integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, n)
        res = integ.reportResults()
        integ.finalize()
        same = (got["fate"] == rf["fate"]) & (got["ix"] == rf["ix"]) & (got["iy"] == rf["iy"]) & (got["iz"] == rf["iz"]) & \
            (got["nScatter"] == rf["nScatter"]) & (np.abs(got["weight"] - rf["weight"]) <= 1e-6)
        assert same[order <= 10].mean() > 0.985, (walk, case["name"], nx, ny, nz, same[order <= 10].mean())
        check_event_totals(walk, case, cnt, got, ref["counters"], rf, same)
        for g, r in ((res["meanFluxUp"], mu), (res["meanFluxDown"], md), (res["meanFluxAbsorbed"], ma)):
            assert abs(g - r) < 1e-2 * scale, (walk, g, r)


@pytest.mark.timeout(120, method="thread")
@pytest.mark.parametrize("seed", range(FUZZ))
def test_random_rich_domains_against_the_oracle(M, seed):
    """The random domains once more with what the plain ones leave out: phase-function tables of several entries with a
    random entry per cell, a third component (no 16-byte collision record for it: the general look-up), a reflecting
    surface of random patches on positions of its own.  Every walk of the product against the oracle, as above."""
    from oracle import oracle as O
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    case, mu0, phi0, rr = random_oracle_case(seed)
    rng = np.random.default_rng(64000 + seed)
    nx, ny, nz = len(case["xe"]) - 1, len(case["ye"]) - 1, len(case["ze"]) - 1
    comps = case["components"]
    if rng.random() < 0.6:  # several table entries for the first component
        ne = int(rng.integers(2, 5))
        comps[0]["legendre"] = [cases.hg_legendre(float(g), 24) for g in rng.uniform(0.0, 0.9, ne)]
        comps[0]["pfIndex"] = rng.integers(1, ne + 1, (nx, ny, nz)).astype(np.int32)
    if rng.random() < 0.6:  # one more 3-D component, sparse, absorbing, its own table
        e3 = rng.uniform(0.0, 10.0, (nx, ny, nz)) * (rng.random((nx, ny, nz)) < 0.3)
        comps.append(dict(ext=e3, ssa=np.where(e3 > 0, rng.uniform(0.3, 1.0, e3.shape), 0.0), pfIndex=rng.integers(1, 4, e3.shape).astype(np.int32),
                          legendre=[cases.hg_legendre(float(g), 16) for g in (0.2, 0.6, 0.85)]))
    if rng.random() < 0.5:
        case = cases.patchy_surface(case, nxs=int(rng.integers(1, 6)), nys=int(rng.integers(1, 5)), seed=int(rng.integers(0, 10 ** 6)))
    n = 15000
    P = cases.oracle_problem(case, nsteps=9001, use_russian_roulette=rr)
    ref = O.compute_rt(P, O.solar_source(mu0, phi0), O.philox_rng(SEED, 0), n, want_fates=True)
    rf, order = ref["fates"], ref["fates"]["nScatter"]
    mu, md, ma, _ = O.report_means(P, O.normalize(P, n, ref))
    for walk, tuning in (("face by face", dict(privateTallies=0, layerSkip=0)), ("layers + flight", dict(privateTallies=0, layerSkip=3)),
                         ("LDS face by face", dict(blockWalk=0)), ("block walk", dict(blockWalk=2))):
        dom = cases.product_domain(case)
        integ = M.new_Integrator(dom)
        integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=rr, surfaceBDRF=cases.product_surface(case))
        integ.setTuning(eventThreshold=16, **tuning)
        photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 9)
        got = integ.traceFates(dom, new_RandomNumberSequence(SEED), photons, n)
        cnt = integ.counters()
        integ.resetMoments()
        integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, n)
        res = integ.reportResults()
        integ.finalize()
        same = (got["fate"] == rf["fate"]) & (got["ix"] == rf["ix"]) & (got["iy"] == rf["iy"]) & (got["iz"] == rf["iz"]) & \
            (got["nScatter"] == rf["nScatter"]) & (np.abs(got["weight"] - rf["weight"]) <= 1e-6)
        assert same[order <= 10].mean() > 0.985, (walk, case["name"], nx, ny, nz, len(comps), same[order <= 10].mean())
        check_event_totals(walk, case, cnt, got, ref["counters"], rf, same)
        for g, r in ((res["meanFluxUp"], mu), (res["meanFluxDown"], md), (res["meanFluxAbsorbed"], ma)):
            assert abs(g - r) < 5e-3, (walk, g, r)


def random_oracle_case(seed):
    rng = np.random.default_rng(31000 + seed)
    nx, ny, nz = 4 * int(rng.integers(1, 4)), 4 * int(rng.integers(1, 3)), int(rng.integers(2, 13))
    def edges(n, stretched, origin):
        d = rng.uniform(0.02, 0.06) * (np.cumprod(rng.uniform(0.85, 1.2, n)) if stretched else np.ones(n))
        return origin + np.concatenate([[0.0], np.cumsum(d)])
    st = rng.random() < 0.4
    xe, ye, ze = edges(nx, st, rng.uniform(-1.0, 1.0) if st else 0.0), edges(ny, st, 0.0), edges(nz, rng.random() < 0.5, 0.0)
    ext = np.zeros((nx, ny, nz))
    for k in range(nz):
        kind = rng.choice(["vacuum", "haze", "broken", "random"])
        if kind == "haze":
            ext[:, :, k] = rng.choice([0.05, 2.0])
        elif kind == "broken":
            ext[:, :, k] = np.where(rng.random((nx, ny)) < rng.uniform(0.1, 0.9), rng.uniform(2.0, 30.0, (nx, ny)), rng.choice([0.0, 0.1]))
        elif kind == "random":
            ext[:, :, k] = rng.uniform(0.0, 20.0, (nx, ny)) * (rng.random((nx, ny)) < 0.9)
    comps = [dict(ext=ext, ssa=np.where(ext > 0, float(rng.uniform(0.6, 1.0)), 0.0), pfIndex=np.ones(ext.shape, np.int32),
                  legendre=[cases.hg_legendre(float(rng.uniform(0.0, 0.9)), 24)])]
    if rng.random() < 0.4:
        comps.append(dict(ext=rng.uniform(0.0, 0.3, nz), ssa=np.ones(nz), pfIndex=np.ones(nz, np.int32),
                          legendre=[np.array([0.0, 0.1], np.float32)]))
    case = dict(name="oracle%d" % seed, xe=xe, ye=ye, ze=ze, albedo=float(rng.choice([0.0, 0.3, 0.8])), components=comps)
    mu0, phi0 = float(rng.choice([1.0, rng.uniform(0.05, 1.0)])), float(rng.uniform(0.0, 360.0))
    rr = bool(rng.integers(0, 2))
    return case, mu0, phi0, rr
