"""Spectrally integrated runs with every wavelength's optics resident on the device (mcbrat3d_amd.broadband.SpectralRun):
thermal (config 4's mechanism) and solar (solar_Weighting), photons split over wavelengths on the device
(getFrequencyDistr), against the same loop over the oracle.  Run on the MI355X box with `-m gpu`."""
import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu
SEED = 1717


@pytest.fixture(scope="module")
def M():
    import mcbrat3d_amd
    return mcbrat3d_amd


def test_frequency_distribution_on_the_device_is_the_oracles(M):
    """getFrequencyDistr (emissionAndBroadBandWeights.f95:552-572): one uniform per photon against the power CDF.  The
    device draws are Philox; the oracle in Philox mode must count the very same photons (exact), and the oracle in
    MT mode -- the reference's generator -- the same distribution (5 sigma per wavelength)."""
    from oracle import oracle as O
    dom = cases.product_domain(cases.plane_parallel())
    integ = M.new_Integrator(dom)
    rs = np.random.default_rng(4)
    for n in (1, 5, 16, 9000):  # (9000 > 8192: the histogram leaves LDS)
        p = rs.random(n) + 0.05
        if n > 1:
            p[rs.integers(0, n)] = 0.0  # a wavelength without power gets no photon
        cdf = np.cumsum(p) / p.sum()
        cdf[-1] = 1.0
        for total, first in ((0, 0), (1, 0), (999, 3), (2 * 10 ** 6, 0), (10 ** 6 + 7, 10 ** 9 + 1)):
            got = integ.frequencyDistribution(cdf, total, seed=SEED, firstDraw=first)
            ref = O.frequency_distribution(O.philox_rng(SEED), cdf, total, first)
            assert got.sum() == total and np.array_equal(got, ref), (n, total, first)
        total = 2 * 10 ** 6
        got = integ.frequencyDistribution(cdf, total, seed=SEED)
        mt = O.frequency_distribution(O.mt_rng([10, 1, 0]), cdf, total)
        pr = np.diff(np.concatenate([[0.0], cdf]))
        assert np.all(np.abs(got - mt) < 5 * np.sqrt(2 * pr * (1 - pr) * total) + 1)
        assert np.all(got[pr == 0.0] == 0)
    # 1e9 photons (a production run's split) in well under a second, against the expectation
    cdf = np.linspace(1 / 16, 1.0, 16)
    got = integ.frequencyDistribution(cdf, 10 ** 9, seed=1)
    assert got.sum() == 10 ** 9 and np.all(np.abs(got - 10 ** 9 / 16) < 6 * np.sqrt(10 ** 9 / 16))
    integ.finalize()


def _oracle_loop(problems, sources, counts, ppb, seed, total_flux):
    from oracle import oracle as O
    batches, first = [], 0
    for P, src, n in zip(problems, sources, counts):
        left = int(n)
        while left > 0:
            k = min(ppb, left)
            r = O.compute_radiative_transfer(P, src, O.philox_rng(seed, first), k)
            batches.append((k, np.array([r["meanFluxUp"], r["meanFluxDown"], r["meanFluxAbsorbed"]], np.float64)))
            first += k
            left -= k
    return O.batch_statistics(batches, solar_flux=total_flux), len(batches)


def test_resident_thermal_spectrum_matches_the_oracle_loop(M):
    from mcbrat3d_amd import broadband, driver
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    from oracle import oracle as O
    lambdas = [8.0, 9.0, 10.5, 11.0, 12.0]
    ppb, nb = 4000, 12
    cs = [cases.homog_lw(n=20, lam=lam, ext=5.0 + 0.5 * (lam - 8.0), ssa=0.5) for lam in lambdas]
    doms = [cases.product_domain(c) for c in cs]
    run = broadband.SpectralRun(M, doms, minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=True)
    flux = run.prepare_thermal(300.0)
    run.resetMoments()
    rng = new_RandomNumberSequence(SEED)
    counts = run.run(ppb, nb, rng, seed=3)
    assert counts.sum() == ppb * nb and rng.nextPhotonId == ppb * nb
    st = driver.statistics(driver.unpack_moments(run.moments(), 20, 20, 20), solarFlux=flux)
    # a second run re-uses the resident domains: nothing is uploaded, results are bitwise the same
    run.resetMoments()
    counts2 = run.run(ppb, nb, new_RandomNumberSequence(SEED), seed=3)
    assert np.array_equal(counts, counts2)
    st2 = driver.statistics(driver.unpack_moments(run.moments(), 20, 20, 20), solarFlux=flux)
    assert st2["meanFluxUp"] == st["meanFluxUp"] and np.array_equal(st2["absorbedVolume"], st["absorbedVolume"])
    # oracle: same widths, emitted power, CDF, photon split, photons
    widths = broadband.spectral_widths(lambdas)
    fl, srcs, probs = [], [], []
    for c, dl in zip(cs, widths):
        P = cases.oracle_problem(c, nsteps=9001, lw_flag=1.0)
        vw, frac, f = O.emission_weighting(P, c["temps"].transpose(2, 1, 0).reshape(-1), c["lambda_um"], 300.0, dl)
        fl.append(f); srcs.append(O.EmissionSource(vw, frac)); probs.append(P)
    cdf, total = broadband.emitted_flux_cdf(fl)
    assert total == pytest.approx(flux, rel=1e-12)
    assert np.array_equal(counts, O.frequency_distribution(O.philox_rng(3), cdf, ppb * nb))
    (mean, err), nbatches = _oracle_loop(probs, srcs, counts, ppb, SEED, total)
    got = np.array([st["meanFluxUp"], st["meanFluxDown"], st["meanFluxAbsorbed"]])
    assert np.all(np.abs(got - mean) < 2e-3 * total), (got, mean)
    assert st["batches"] == nbatches
    assert st["meanFluxAbsorbed"] < 0 < st["meanFluxUp"]
    # a caller-owned moment buffer (what a multi-GPU run all-reduces): every wavelength's context must follow it
    import torch
    buf = torch.zeros(8 + 2 * run.first.momentsLength(), dtype=torch.float64, device="cuda:0")
    run.bindMoments(buf.data_ptr())
    run.resetMoments()
    run.run(ppb, nb, new_RandomNumberSequence(SEED), seed=3)
    torch.cuda.synchronize()
    st3 = driver.statistics(driver.unpack_moments(buf.cpu().numpy(), 20, 20, 20), solarFlux=flux)
    assert st3["totalPhotons"] == ppb * nb and st3["meanFluxUp"] == st["meanFluxUp"]
    assert np.array_equal(st3["absorbedVolume"], st["absorbedVolume"])
    run.finalize()


def test_overlapping_wavelengths_are_bitwise_the_host_serialised_run(M):
    """SpectralRun(overlap=True): every wavelength's integrator asynchronous, the finish chains of consecutive calls ordered on
    the device (mcbrat_chain_after), tracing kernels of different wavelengths overlapping -- against overlap=False, where the
    host waits for every call: the same photons folded in the same order, so the moment arrays must be bitwise equal; twice
    in a row (a second run re-uses lanes and events), with full batches and a rest batch per wavelength."""
    from mcbrat3d_amd import broadband
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    lambdas = [8.0, 9.5, 10.5, 12.0]
    cs = [cases.homog_lw(n=20, lam=lam, ext=4.0 + lam / 4.0, ssa=0.6) for lam in lambdas]
    out = {}
    for overlap in (False, True):
        run = broadband.SpectralRun(M, [cases.product_domain(c) for c in cs], overlap=overlap, minInverseTableSize=9001)
        run.prepare_thermal(300.0)
        res = []
        for rep in range(2):
            run.resetMoments()
            counts = run.run(7000, 9, new_RandomNumberSequence(SEED + rep), seed=11 + rep)
            res.append((counts.copy(), run.moments().copy()))
        assert sum(it.badPhotons() for it in run.integrators) == 0
        run.finalize()
        out[overlap] = res
    for a, b in zip(out[False], out[True]):
        assert np.array_equal(a[0], b[0]) and a[1][0] == 63000
        assert np.array_equal(a[1], b[1])


def test_resident_solar_spectrum_matches_the_oracle_loop(M):
    """Solar broadband: solar_Weighting (emissionAndBroadBandWeights.f95:149-217) over four wavelength domains of the
    step cloud whose extinction and single-scattering albedo change with wavelength."""
    from mcbrat3d_amd import broadband, driver
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    from oracle import oracle as O
    lambdas = np.array([0.45, 0.65, 0.85, 1.6])
    source = np.array([1.9, 1.5, 1.0, 0.25])  # W m^-2 um^-1
    ssas = [1.0, 0.999, 0.99, 0.9]
    mu0, phi0 = 0.6, 20.0
    ppb, nb = 5000, 10
    cs = []
    for lam, w in zip(lambdas, ssas):
        c = cases.step_cloud(ssa=w)
        c["components"][0]["ext"] = c["components"][0]["ext"] * (0.65 / lam) ** 0.2
        c["albedo"] = 0.2
        c["lambda_um"] = float(lam)
        cs.append(c)
    doms = [cases.product_domain(c) for c in cs]
    run = broadband.SpectralRun(M, doms, minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=True)
    flux = run.prepare_solar(mu0, phi0, source)
    cdf, total = broadband.solar_weighting(source, lambdas, mu0)
    assert flux == total and cdf[-1] == 1.0
    # solar_Weighting by hand: dLambda x mu0 x S, half-way widths inside, one-sided at both ends
    w = np.array([0.2, 0.2, 0.475, 0.75]) * np.float32(mu0) * source
    assert np.allclose(np.cumsum(w) / w.sum(), cdf, rtol=1e-14) and total == pytest.approx(w.sum(), rel=1e-14)
    run.resetMoments()
    counts = run.run(ppb, nb, new_RandomNumberSequence(SEED), seed=8)
    st = driver.statistics(driver.unpack_moments(run.moments(), 32, 1, 32), solarFlux=flux)
    probs = [cases.oracle_problem(c, nsteps=9001) for c in cs]
    assert np.array_equal(counts, O.frequency_distribution(O.philox_rng(8), cdf, ppb * nb))
    (mean, err), nbatches = _oracle_loop(probs, [O.solar_source(mu0, phi0)] * 4, counts, ppb, SEED, total)
    got = np.array([st["meanFluxUp"], st["meanFluxDown"], st["meanFluxAbsorbed"]])
    assert np.all(np.abs(got - mean) < 2e-3 * total), (got, mean)
    assert st["batches"] == nbatches and st["totalPhotons"] == ppb * nb
    run.finalize()


def test_driver_cli_runs_the_ssp_file_family(M, tmp_path):
    """The Python driver on the file family the reference's current driver reads: physDomainFile + two SSP table files +
    solarSourceFile, three wavelengths -> a spectrally integrated solar run; the same numbers as SpectralRun driven by
    hand on the domains the readers return (same seed: bitwise), and the NetCDF result file in the reference's shape."""
    from scipy.io import netcdf_file
    from mcbrat3d_amd import broadband, driver, driver_cli, ncio
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    from tests import test_ncio_ssp as T
    rng = np.random.default_rng(11)
    T.write_common(str(tmp_path / "phys.nc"), rng)
    key = np.array([5.0, 10.0, 15.0, 20.0], np.float32)
    liquid = dict(name="liquid", zLevelBase=1, extType="volExt", key=key, extT=0.5 + rng.random((4, T.NLAM)),
                  ssaT=0.9 + 0.1 * rng.random((4, T.NLAM)),
                  legendre=[[np.array([np.float32(0.8) ** n for n in range(1, 9 + k)], np.float32) for k in range(4)] for _ in range(T.NLAM)])
    ice = dict(liquid, name="ice", extT=0.2 + rng.random((4, T.NLAM)))
    albedo = np.array([0.05, 0.2, 0.35])
    T.write_ssp(str(tmp_path / "ssp1.nc"), [liquid], albedo)
    T.write_ssp(str(tmp_path / "ssp2.nc"), [ice], albedo)
    nml = T.write_run_files(tmp_path, None)
    st = driver_cli.main([str(nml)])
    assert st["totalPhotons"] == 18000
    cfg = driver_cli.read_namelists(str(nml))
    doms = driver_cli.load_domains(cfg)
    run = broadband.SpectralRun(M, doms, minInverseTableSize=9001)
    src, lam = ncio.read_SolarSource(str(tmp_path / "sun.nc"), 3)
    flux = run.prepare_solar(0.7, 15.0, src, lam)
    run.resetMoments()
    run.run(3000, 6, new_RandomNumberSequence(21), seed=21)
    want = driver.statistics(driver.unpack_moments(run.moments(), T.NX, T.NY, T.NZ), solarFlux=flux)
    run.finalize()
    for k in ("meanFluxUp", "meanFluxDown", "meanFluxAbsorbed", "meanFluxUp_StdErr"):
        assert st[k] == want[k], k
    assert np.array_equal(st["fluxDown"], want["fluxDown"]) and np.array_equal(st["absorbedProfile"], want["absorbedProfile"])
    assert 0 < st["meanFluxUp"] < flux and st["meanFluxDown"] > 0
    f = netcdf_file(str(tmp_path / "out.nc"), "r", mmap=False)
    assert f.variables["fluxUp"].shape == (T.NY, T.NX) and f.variables["absorptionProfile"].shape == (T.NZ,)
    assert f.Solar_flux == pytest.approx(flux) and np.allclose(f.variables["fluxUp"][:].T, st["fluxUp"], rtol=1e-6)
    f.close()
    assert "Flux_Up" in open(tmp_path / "flux.out").read()
