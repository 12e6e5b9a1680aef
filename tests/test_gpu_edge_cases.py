"""Edge cases of the domain, through the C ABI on the GPU: tiny and ragged photon counts, more
batches than workgroups, reflecting surface, vacuum, grazing sun, roulette off, private vs
global tallies, and size-independent properties at BASELINE.json's full photon counts."""
import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu
SEED = 31337


@pytest.fixture(scope="module")
def M():
    import mcbrat3d_amd
    return mcbrat3d_amd


def _run(M, case, mu0, phi0, ppb, nb, rr=True, seed=SEED, tuning=None, first=0):
    from mcbrat3d_amd import driver
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=rr)
    if tuning:
        integ.setTuning(**tuning)
    photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 12)
    integ.resetMoments()
    n = integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(seed, first), photons, ppb, nb)
    mom = integ.moments()
    st = driver.statistics(driver.unpack_moments(mom, dom.numX, dom.numY, dom.numZ))
    last = integ.reportResults()
    integ.finalize()
    return n, st, last, mom


def _oracle(case, mu0, phi0, n, rr=True, seed=SEED, first=0):
    from oracle import oracle as O
    P = cases.oracle_problem(case, nsteps=9001, use_russian_roulette=rr)
    return O.compute_radiative_transfer(P, O.solar_source(mu0, phi0), O.philox_rng(seed, first), n)


@pytest.mark.parametrize("n", [1, 63, 65, 257, 1000])
def test_tiny_and_ragged_photon_counts(M, n):
    """Fewer photons than lanes in a wave / not a multiple of anything."""
    case = cases.step_cloud(0.99)
    done, st, last, _ = _run(M, case, 1.0, 0.0, n, 1)
    ref = _oracle(case, 1.0, 0.0, n)
    assert done == n and st["totalPhotons"] == n
    assert np.allclose(last["fluxUp"][:, 0], ref["fluxUp"], atol=32.0 / n * 1.01 * (n > 200) + 1e-5)
    assert abs(last["meanFluxUp"] + last["meanFluxDown"] + last["meanFluxAbsorbed"] - 1.0) < 0.05 + 2.0 / n


@pytest.mark.parametrize("first", [2 ** 32 - 700, 2 ** 40 + 12345])
def test_photon_ids_beyond_32_bits(M, first):
    """A production run of more than 4.3e9 photons crosses 2^32 photon ids: the id is the Philox counter's low and high word
    (getRandomReal's stream position in the reference: one stream per process, RandomNumbersForMC.f95:277-292).  A batch
    that straddles the boundary, and one far beyond it, against the oracle started at the same id."""
    case = cases.step_cloud(0.99)
    n = 2000
    done, st, last, _ = _run(M, case, 1.0, 0.0, n, 1, first=first)
    ref = _oracle(case, 1.0, 0.0, n, first=first)
    assert done == n and st["totalPhotons"] == n
    assert np.allclose(last["fluxUp"][:, 0], ref["fluxUp"], atol=32.0 / n * 2.01 + 1e-5)   # (two flipped histories at most)
    assert np.allclose(last["fluxDown"][:, 0], ref["fluxDown"], atol=32.0 / n * 2.01 + 1e-5)
    # ... and they are other photons than those of ids 0 .. n-1
    base = _run(M, case, 1.0, 0.0, n, 1, first=0)[2]
    assert not np.array_equal(base["fluxUp"], last["fluxUp"])


def test_more_batches_than_workgroups(M):
    """3000 batches of 512 photons: workgroups loop over units; moments fold every batch."""
    case = cases.plane_parallel(ssa=0.9)
    done, st, _, _ = _run(M, case, 1.0, 0.0, 512, 3000)
    assert done == 512 * 3000 and st["batches"] == 3000
    ref = _oracle(case, 1.0, 0.0, 200000)
    for k in ("meanFluxUp", "meanFluxDown", "meanFluxAbsorbed"):
        assert abs(st[k] - ref[k]) < 5 * np.sqrt(st[k + "_StdErr"] ** 2 + 0.25 / 200000) + 1e-4, k
    assert 0 < st["meanFluxUp_StdErr"] < 2e-3


def test_private_and_global_tallies_agree_bitwise(M):
    """LDS-private slabs (small domains) and global atomics are both exact integer sums."""
    case = cases.step_cloud(0.99)
    _, _, _, a = _run(M, case, 0.5, 30.0, 20000, 7, tuning=dict(privateTallies=1, blockWalk=0))
    _, _, _, b = _run(M, case, 0.5, 30.0, 20000, 7, tuning=dict(privateTallies=0, blockSize=256, eventThreshold=40))
    assert np.array_equal(a, b)


def test_reflecting_surface_and_no_roulette(M):
    """Lambertian albedo 0.6: photons bounce off the surface repeatedly (computeRT :619-676)."""
    case = cases.step_cloud(1.0)
    case["albedo"] = 0.6
    n = 60000
    done, st, last, _ = _run(M, case, 0.5, 0.0, n, 1, rr=False)
    ref = _oracle(case, 0.5, 0.0, n, rr=False)
    for k in ("meanFluxUp", "meanFluxDown", "meanFluxAbsorbed"):
        assert abs(last[k] - ref[k]) < 30.0 / n, (k, last[k], ref[k])
    assert last["meanFluxDown"] > 0.5 and last["meanFluxAbsorbed"] == 0.0
    # conservative medium: what comes down and is not reflected is absorbed by the surface
    assert abs(last["meanFluxUp"] + (1 - 0.6) * last["meanFluxDown"] - 1.0) < 2e-3


def test_vacuum_and_grazing_sun(M):
    """Zero extinction everywhere: every photon reaches the surface in its launch column,
    after wrapping around the periodic domain many times for a grazing sun."""
    case = cases.step_cloud(1.0)
    case["components"][0]["ext"][:] = 0.0
    for mu0 in (1.0, 0.02):
        n = 20000
        done, st, last, _ = _run(M, case, mu0, 45.0, n, 1)
        assert done == n
        assert last["meanFluxDown"] == pytest.approx(1.0, abs=1e-6)
        assert last["meanFluxUp"] == 0.0 and last["meanFluxAbsorbed"] == 0.0
        assert np.all(last["volumeAbsorption"] == 0.0)


def test_full_size_properties_step_cloud(M):
    """BASELINE.json configs[1] at full size (1e7 photons) and the accuracy target's size (1e8):
    properties that need no reference run -- energy closure, mean of the batch means equals the
    whole, column sums equal domain means, independence of how the batches are cut."""
    case = cases.step_cloud(0.99)
    n1, st1, _, m1 = _run(M, case, 1.0, 0.0, 100000, 100)
    n2, st2, _, m2 = _run(M, case, 1.0, 0.0, 1000000, 10)
    assert n1 == n2 == 10 ** 7
    closure = st1["meanFluxUp"] + st1["meanFluxDown"] + st1["meanFluxAbsorbed"]
    assert abs(closure - 1.0) < 3.0 / np.sqrt(n1)  # roulette leaves closure only statistically exact
    for k in ("meanFluxUp", "meanFluxDown", "meanFluxAbsorbed"):
        assert abs(st1[k] - st2[k]) < 1e-6  # same photons, different batching: float32 normalisation only
        assert abs(st1[k.replace("meanF", "f")].mean() - st1[k]) < 1e-6
    assert abs(st1["absorbedProfile"].sum() * 0.0078125 * 1000.0 - st1["meanFluxAbsorbed"]) < 1e-5
    n3, st3, _, _ = _run(M, case, 1.0, 0.0, 1000000, 100, seed=2)
    assert n3 == 10 ** 8
    assert abs(st3["meanFluxUp"] + st3["meanFluxDown"] + st3["meanFluxAbsorbed"] - 1.0) < 3.0 / np.sqrt(n3)
    for k in ("meanFluxUp", "meanFluxDown", "meanFluxAbsorbed"):
        assert abs(st3[k] - st1[k]) < 5 * np.sqrt(st3[k + "_StdErr"] ** 2 + st1[k + "_StdErr"] ** 2)


def test_error_paths_through_the_abi(M):
    case = cases.step_cloud(0.99)
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    with pytest.raises(M.McbratError, match="useRayTracing"):
        integ.specifyParameters(useRayTracing=False)
    integ.useRayTracing = True
    with pytest.raises(M.McbratError, match="intensity"):
        integ.specifyParameters(computeIntensity=True)
    with pytest.raises(M.McbratError, match="no batch"):
        integ.reportResults()
    photons = M.new_PhotonStream(1.0, 0.0, numberOfPhotons=0)
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    with pytest.raises(M.McbratError, match="Didn't process any photons"):
        integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(1), photons, 1000)
    integ.finalize()


def test_brick_layout_is_lossless(M):
    """4x4x4 bricks with unstored background bricks vs the dense grids: the kernel must read the
    same floats, so the moment arrays are bitwise equal -- on a cloud field with a large clear
    fraction (two components) and on a domain without any background (all bricks stored, with
    ragged brick edges: 30 x 3 x 13 cells)."""
    case = cases.landsat_like(n=48, nz=24, n_entries=6)
    _, _, _, a = _run(M, case, 0.5, 30.0, 30000, 4, tuning=dict(brickLayout=0, eventThreshold=32, layerSkip=0))
    _, _, _, b = _run(M, case, 0.5, 30.0, 30000, 4, tuning=dict(brickLayout=1, eventThreshold=32))
    assert np.array_equal(a, b)  # (the brick walk stops at every face: compare it with the dense walk that does so too)
    rng = np.random.default_rng(5)
    ext = rng.uniform(0.5, 40.0, (30, 3, 13))
    odd = dict(name="odd", xe=0.01 * np.arange(31), ye=0.02 * np.arange(4), ze=np.concatenate([[0.0], np.cumsum(rng.uniform(0.01, 0.03, 13))]),
               components=[dict(ext=ext, ssa=rng.uniform(0.8, 1.0, ext.shape), pfIndex=rng.integers(1, 4, ext.shape).astype(np.int32),
                                legendre=[cases.hg_legendre(g, 16) for g in (0.6, 0.75, 0.85)])], albedo=0.3)
    _, _, _, a = _run(M, odd, 0.7, 120.0, 20000, 3, tuning=dict(brickLayout=0, privateTallies=0))
    _, _, _, b = _run(M, odd, 0.7, 120.0, 20000, 3, tuning=dict(brickLayout=1, privateTallies=0))
    assert np.array_equal(a, b)
    ref = _oracle(odd, 0.7, 120.0, 20000)
    _, _, last, _ = _run(M, odd, 0.7, 120.0, 20000, 1, tuning=dict(brickLayout=1))
    for k in ("meanFluxUp", "meanFluxDown", "meanFluxAbsorbed"):
        assert abs(last[k] - ref[k]) < 2e-3, (k, last[k], ref[k])


def test_async_calls_overlap_and_match_synchronous_mode(M):
    """The reference's driver calls computeRadiativeTransfer once per batch (monteCarloDriver.f95:1008).
    In asynchronous mode those calls overlap on the GPU; moments are folded in call order, so they must
    be bitwise what the synchronous calls give -- also with a reset in between and a ragged last call."""
    import time
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    case = cases.step_cloud(0.99)
    dom = cases.product_domain(case)
    photons = M.new_PhotonStream(1.0, 0.0, numberOfPhotons=10 ** 12)
    out = {}
    for mode in ("sync", "async", "async_auto"):
        integ = M.new_Integrator(dom)
        integ.specifyParameters(minInverseTableSize=9001)
        if mode != "async_auto":  # (default: the library picks the threshold; calls this small get its guess, 16 here)
            integ.setTuning(eventThreshold=16)
        integ.setAsync(mode != "sync")
        rng = new_RandomNumberSequence(SEED)
        integ.resetMoments()
        integ.computeRadiativeTransfer(dom, rng, photons, 20000, 3)   # discarded by the reset below
        integ.resetMoments()
        integ.synchronize()
        t0 = time.time()
        for k in range(40):
            integ.computeRadiativeTransfer(dom, rng, photons, 50000 if k < 39 else 12345, 1)
        integ.synchronize()
        dt = time.time() - t0
        out[mode] = (integ.moments(), integ.reportResults(), dt, integ.lastTraceMs())
        integ.finalize()
    ms, rs, ts, ks = out["sync"]
    ma, ra, ta, ka = out["async"]
    assert np.array_equal(ms, out["async_auto"][0]) and out["async_auto"][2] < 1.5 * ts
    assert ms[0] == ma[0] == 39 * 50000 + 12345 and ms[1] == ma[1] == 40
    assert np.array_equal(ms, ma)
    for k in rs:
        assert np.array_equal(np.asarray(rs[k]), np.asarray(ra[k])), k
    assert ka > 0.0
    print("40 per-batch calls: synchronous %.1f ms, asynchronous %.1f ms (kernel time %.1f / %.1f ms)" % (ts * 1e3, ta * 1e3, ks, ka))
    assert ta < 1.5 * ts  # normally about half (the drain of one call overlaps the next); loose: wall clocks on a shared box


def test_surface_description_patches(M):
    """specifyParameters(surfaceBDRF=): a reflecting surface of 5 x 3 patches (one of them black) on positions that
    differ from the grid's, under the step cloud and under a broken cloud field; per-photon fates and the batch's
    fluxes against the oracle on the same Philox streams, and the domain's own albedo must be ignored."""
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    from oracle import oracle as O
    for base, mu0, phi0 in ((cases.step_cloud(0.99), 0.7, 20.0), (cases.landsat_like(n=48, nz=24, n_entries=6), 0.5, 30.0)):
        case = cases.patchy_surface(base)
        case["albedo"] = 0.77  # ignored once a surface description is given (computeRT :667-673)
        n = 30000
        dom = cases.product_domain(case)
        integ = M.new_Integrator(dom)
        integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=True,
                                surfaceBDRF=cases.product_surface(case))
        photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 9)
        got = integ.traceFates(dom, new_RandomNumberSequence(SEED), photons, n)
        integ.resetMoments()
        integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, n)
        res = integ.reportResults()
        integ.finalize()
        P = cases.oracle_problem(case, nsteps=9001)
        ref = O.compute_rt(P, O.solar_source(mu0, phi0), O.philox_rng(SEED, 0), n, want_fates=True)
        rf = ref["fates"]
        same = (got["fate"] == rf["fate"]) & (got["ix"] == rf["ix"]) & (got["iy"] == rf["iy"]) & \
            (got["nScatter"] == rf["nScatter"]) & (np.abs(got["weight"] - rf["weight"]) <= 1e-6)
        assert same.mean() > 0.99, "only %.4f of photon histories identical" % same.mean()
        assert (rf["fate"] == 1).sum() > 50  # photons that met the black patch
        mu, md, ma, _ = O.report_means(P, O.normalize(P, n, ref))
        for g, r in ((res["meanFluxUp"], mu), (res["meanFluxDown"], md), (res["meanFluxAbsorbed"], ma)):
            assert abs(g - r) < 3e-3, (g, r)
    # copy_Integrator (:1296-1376): same domain, same parameters (surface description included) -> same results
    dom = cases.product_domain(case)
    first = M.new_Integrator(dom)
    first.specifyParameters(minInverseTableSize=9001, surfaceBDRF=cases.product_surface(case))
    second = first.copy_Integrator()
    out = []
    for integ in (first, second):
        integ.resetMoments()
        integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 9), 5000)
        out.append(integ.moments().copy())
        integ.finalize()
    assert np.array_equal(out[0], out[1])
    with pytest.raises(M.McbratError, match="surface description isn't valid"):
        integ2 = M.new_Integrator(cases.product_domain(cases.step_cloud(0.99)))
        integ2.specifyParameters(surfaceBDRF=object())


def test_very_tall_grid(M):
    """3000 layers on 2 x 2 columns: the edge and per-layer tables (72 KB) pass the default LDS limit of a workgroup."""
    nz = 3000
    ze = np.linspace(0.0, 3.0, nz + 1)
    ext = np.full((2, 2, nz), 1.5)
    ext[:, :, 1000:2000] = 0.0          # a run of 1000 clear layers in the middle
    ext[0, 0, ::7] = 3.0                # and layers with cell-to-cell extinction below and above it
    ext[0, 0, 1000:2000] = 0.0
    case = dict(name="tall", xe=np.array([0.0, 0.1, 0.2]), ye=np.array([0.0, 0.1, 0.2]), ze=ze, albedo=0.2,
                components=[dict(ext=ext, ssa=np.where(ext > 0, 0.9, 0.0), pfIndex=np.ones(ext.shape, np.int32),
                                 legendre=[cases.hg_legendre(0.7, 16)])])
    n = 20000
    done, st, last, _ = _run(M, case, 0.8, 40.0, n, 1)
    ref = _oracle(case, 0.8, 40.0, n)
    assert done == n
    for k in ("meanFluxUp", "meanFluxDown", "meanFluxAbsorbed"):
        assert abs(last[k] - ref[k]) < 3e-3, (k, last[k], ref[k])


def test_one_integrator_reused_with_temporary_streams_and_changed_domains(M):
    """The integrator remembers what is on the device by content, not by object identity: CPython hands the id of a
    freed temporary to the next one, so three `new_PhotonStream(mu0, ...)` temporaries in a loop used to look like
    one stream.  Same for a changed surface albedo and an added component."""
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    case = cases.step_cloud(0.99)
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=9001)
    n = 200000
    downs = []
    for mu0 in (1.0, 0.5, 0.25):
        integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(5), M.new_PhotonStream(mu0, 0.0, numberOfPhotons=n), n)
        downs.append(integ.reportResults()["meanFluxDown"])
    fresh = []
    for mu0 in (1.0, 0.5, 0.25):
        _, _, last, _ = _run(M, case, mu0, 0.0, n, 1, seed=5)
        fresh.append(last["meanFluxDown"])
    assert downs == fresh  # bitwise what a fresh integrator per sun angle gives
    assert downs[0] > downs[1] > downs[2]
    # a mutated stream object
    ps = M.new_PhotonStream(1.0, 0.0, numberOfPhotons=10 * n)
    integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(5), ps, n)
    ps.solarMu = 0.25
    integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(5), ps, n)
    assert integ.reportResults()["meanFluxDown"] == fresh[2]
    # the domain's albedo changed in place: photons now come back from the surface
    up0 = integ.reportResults()["meanFluxUp"]
    dom.surfaceAlbedo = 0.8
    integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(5), ps, n)
    assert integ.reportResults()["meanFluxUp"] > up0 + 0.1
    integ.finalize()


def test_one_context_reconfigured_through_the_abi(M):
    """include/mcbrat.h allows set_grid / set_optics to be called again on a live context: buffers sized by the earlier
    configuration (last-batch results, batch slabs whose stride grows with the number of components when
    limitIntensityContributions is on) must be re-sized, not reused.  Small grid, then a larger one,
    then one -> two components with clipped radiance contributions; each against a fresh context, bitwise."""
    import ctypes as C
    from mcbrat3d_amd._capi import check, lib, ptr
    from oracle import oracle as O  # (only to expand the components into the arrays the ABI takes)
    L = lib()

    def configure(ctx, case, inten, regrid=True):
        nx, ny, nz = len(case["xe"]) - 1, len(case["ye"]) - 1, len(case["ze"]) - 1
        tot, cum, ssa, pfi = O.optical_properties_by_component(nx, ny, nz, case["components"])
        if regrid:
            check(ctx, L.mcbrat_set_grid(ctx, nx, ny, nz, ptr(np.ascontiguousarray(case["xe"], np.float64)),
                                         ptr(np.ascontiguousarray(case["ye"], np.float64)), ptr(np.ascontiguousarray(case["ze"], np.float64))))
        nc = len(case["components"])
        check(ctx, L.mcbrat_set_optics(ctx, nc, ptr(tot), ptr(cum), ptr(ssa), ptr(pfi), C.c_double(case["albedo"])))
        dom = cases.product_domain(case)
        for c, t in enumerate(dom.tabulateInversePhaseFunctions(9001)):
            t = np.ascontiguousarray(t, np.float32)
            check(ctx, L.mcbrat_set_inverse_table(ctx, c + 1, t.shape[1], t.shape[0], ptr(t)))
        mus, phis = np.array([1.0, 0.6], np.float32), np.array([0.0, 40.0], np.float32)
        if inten:
            tab, _ = dom.tabulateForwardPhaseFunctions(9001)
            check(ctx, L.mcbrat_specify_intensity(ctx, 2, ptr(mus), ptr(phis), 0, C.c_float(0.3), 0, 0, 1, C.c_float(0.05)))
            for c, t in enumerate(tab):
                t = np.ascontiguousarray(t, np.float32)
                check(ctx, L.mcbrat_set_forward_table(ctx, c + 1, t.shape[1], t.shape[0], ptr(t), None))
        else:
            check(ctx, L.mcbrat_specify_intensity(ctx, 0, None, None, 0, C.c_float(0.3), 0, 0, 0, C.c_float(1e30)))
        check(ctx, L.mcbrat_specify_parameters(ctx, 1, 1, C.c_float(-1.0)))
        check(ctx, L.mcbrat_set_source_solar(ctx, C.c_float(0.7), C.c_float(20.0)))
        return nx * ny, nz

    def trace(ctx):
        done = C.c_int64(0)
        check(ctx, L.mcbrat_reset_moments(ctx))
        check(ctx, L.mcbrat_compute_radiative_transfer(ctx, 9, 0, 3000, 7, C.byref(done)))
        buf = np.zeros(8 + 2 * int(L.mcbrat_moments_length(ctx)), np.float64)
        check(ctx, L.mcbrat_get_moments(ctx, ptr(buf)))
        return buf

    small = cases.plane_parallel(ssa=0.9)
    big = cases.stretched_grid_cloud()
    one = cases.landsat_like(n=12, nz=10, rayleigh=False)
    two = cases.landsat_like(n=12, nz=10, rayleigh=True)
    steps = [(small, False), (big, False), (one, True), (two, True), (small, False)]
    ctx = L.mcbrat_create(0)
    assert ctx
    reused = []
    for i, (case, inten) in enumerate(steps):
        configure(ctx, case, inten, regrid=(i != 3))  # one -> two components on the SAME grid: set_optics alone
        reused.append(trace(ctx))
    L.mcbrat_destroy(ctx)
    for (case, inten), got in zip(steps, reused):
        fresh = L.mcbrat_create(0)
        configure(fresh, case, inten)
        want = trace(fresh)
        L.mcbrat_destroy(fresh)
        assert got.shape == want.shape and np.array_equal(got, want), case["name"]
