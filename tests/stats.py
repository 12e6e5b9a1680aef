"""The parity statistic of SURVEY.md section 8d, shared by the CPU and GPU tests (test infrastructure).

    z = (A - B) / sqrt(sigma_A^2 + sigma_B^2),  sigma from the batch variance -- the driver's own estimator
    (Drivers/monteCarloDriver.f95:1188-1219) -- over the three domain means, every column flux and every level of
    the absorption (heating) profile.  Pass: max |z| < max(4, sqrt(2 ln N) + 1) over N bins, |mean z| < 0.2 over the
    column bins, domain means within the stated number of sigma.

oracle_run() traces a case with the CPU oracle in either of its two generator modes, spread over processes:
    "mt"     -- the reference's MT19937 stream and draw order (one stream per process, seeded (/iseed, proc, 0/) and
                carried across batches as the driver does, monteCarloDriver.f95:901): the reference-faithful mode;
    "philox" -- the product's counter-based streams (photon id -> stream) and slot table.
"""
import multiprocessing as mp
import os

import numpy as np

from tests import cases

QUANTITIES = ("means", "columns", "profile")


def _worker(args):
    make, make_kw, mode, mu0, phi0, seed, proc, first_batch, n_batches, ppb, roulette = args
    from oracle import oracle as O
    case = getattr(cases, make)(**make_kw)
    P = cases.oracle_problem(case, use_russian_roulette=roulette)
    src = O.solar_source(mu0, phi0)
    rng = O.mt_rng([seed, proc, 0]) if mode == "mt" else None
    out = []
    for b in range(first_batch, first_batch + n_batches):
        if mode == "philox":
            rng = O.philox_rng(seed, b * ppb)
        r = O.compute_radiative_transfer(P, src, rng, ppb)
        out.append((ppb, np.array([r["meanFluxUp"], r["meanFluxDown"], r["meanFluxAbsorbed"]], np.float64),
                    np.concatenate([r["fluxUp"], r["fluxDown"], r["fluxAbsorbed"]]).astype(np.float64),
                    np.asarray(r["absorbedProfile"], np.float64)))
    return out


def oracle_run(make, make_kw, mode, n_batches, ppb, mu0, phi0, seed=10, procs=None, roulette=True):
    """-> dict quantity -> list of (n, values) per batch (what oracle.batch_statistics takes)."""
    from oracle import oracle as O
    O.build()
    procs = procs or max(1, min(8, len(os.sched_getaffinity(0)), n_batches))
    base, extra = divmod(n_batches, procs)
    jobs, lo = [], 0
    for p in range(procs):
        nb = base + (1 if p < extra else 0)
        if nb:
            jobs.append((make, make_kw, mode, mu0, phi0, seed, p + 1, lo, nb, ppb, roulette))
        lo += nb
    if len(jobs) == 1:
        parts = [_worker(jobs[0])]
    else:
        with mp.get_context("spawn").Pool(len(jobs)) as pool:
            parts = pool.map(_worker, jobs)
    rows = [r for part in parts for r in part]
    return {"means": [(n, a) for n, a, _, _ in rows], "columns": [(n, c) for n, _, c, _ in rows],
            "profile": [(n, p) for n, _, _, p in rows]}


def mean_err(batches):
    from oracle import oracle as O
    return O.batch_statistics(batches)


def gpu_mean_err(stats):
    """driver.statistics() result -> the same three (mean, stderr) pairs, column order [fluxUp | fluxDown | fluxAbsorbed], x fastest."""
    m = np.array([stats["meanFluxUp"], stats["meanFluxDown"], stats["meanFluxAbsorbed"]])
    me = np.array([stats["meanFluxUp_StdErr"], stats["meanFluxDown_StdErr"], stats["meanFluxAbsorbed_StdErr"]])
    c = np.concatenate([stats[k].T.reshape(-1) for k in ("fluxUp", "fluxDown", "fluxAbsorbed")])
    ce = np.concatenate([stats[k + "_StdErr"].T.reshape(-1) for k in ("fluxUp", "fluxDown", "fluxAbsorbed")])
    return {"means": (m, me), "columns": (c, ce),
            "profile": (np.asarray(stats["absorbedProfile"]), np.asarray(stats["absorbedProfile_StdErr"]))}


def z_scores(a, b):
    (ma, ea), (mb, eb) = a, b
    live = (np.asarray(ea) > 0) | (np.asarray(eb) > 0)  # (bins nothing ever lands in carry no statistic)
    z = (np.asarray(ma) - np.asarray(mb)) / np.sqrt(np.asarray(ea) ** 2 + np.asarray(eb) ** 2 + 1e-300)
    return z[live]


def assert_parity(a, b, label, mean_sigma=4.0, bin_sigma=4.0, mean_z=0.2, bin_max=True):
    """a, b: dict quantity -> (mean, stderr).  Thresholds of SURVEY.md section 8d."""
    zm = z_scores(a["means"], b["means"])
    zc = z_scores(a["columns"], b["columns"])
    zp = z_scores(a["profile"], b["profile"])
    report = dict(label=label, z_means=[round(float(x), 2) for x in zm], max_z_col=float(np.max(np.abs(zc))),
                  mean_z_col=float(np.mean(zc)), std_z_col=float(np.std(zc)), n_col=int(zc.size),
                  max_z_level=float(np.max(np.abs(zp))) if zp.size else 0.0,
                  mean_z_level=float(np.mean(zp)) if zp.size else 0.0, n_level=int(zp.size))
    assert np.max(np.abs(zm)) < mean_sigma, report
    # SURVEY.md 8d asks for max |z| < 4 over the bins; the largest of N unit normals grows like sqrt(2 ln N) (3.0 for
    # 96 bins, 4.0 for 3072, 4.65 for 49 152), so beyond a few hundred bins the limit is that value plus one
    lim = max(bin_sigma, np.sqrt(2.0 * np.log(max(zc.size, 2))) + 1.0)
    if bin_max:  # (off where a bin of the smaller sample holds a handful of photons: its estimate is Poisson, not Gaussian)
        assert np.max(np.abs(zc)) < lim, report
    assert abs(np.mean(zc)) < mean_z + 3.0 / np.sqrt(zc.size), report  # (0.2, plus the sampling error of a mean over few bins)
    assert 0.7 < np.std(zc) < 1.3 or zc.size < 30, report
    if zp.size:
        assert np.max(np.abs(zp)) < bin_sigma, report
    return report


# ---- config 4: broadband thermal emission, homogeneous isothermal 20x20x20, 16 wavelengths 8-12 um ----
LW_LAMBDAS = tuple(float(x) for x in np.linspace(8.0, 12.0, 16))
LW_SURFACE_TEMP = 300.0


def lw_cases(lambdas=LW_LAMBDAS, n=20):
    """One case per wavelength domain of config 4 (Domain-Files/homogBBDomain.f95:39-66)."""
    return [cases.homog_lw(n=n, lam=lam) for lam in lambdas]


def _lw_worker(args):
    """The reference's thermal broadband loop with its own generator (the reference-faithful oracle mode): emission
    weighting per wavelength (emissionAndBroadBandWeights.f95:424-550), the power CDF, one MT uniform per photon for its
    wavelength (getFrequencyDistr :552-572), then every wavelength's photons in batches (monteCarloDriver.f95:889-1085)."""
    lambdas, n, seed, proc, total, ppb = args
    import time
    from oracle import oracle as O
    from mcbrat3d_amd import broadband
    t0 = time.time()
    cs = lw_cases(lambdas, n)
    widths = broadband.spectral_widths(list(lambdas))
    probs, srcs, fl = [], [], []
    for c, dl in zip(cs, widths):
        P = cases.oracle_problem(c, nsteps=9001, lw_flag=1.0)
        vw, frac, f = O.emission_weighting(P, c["temps"].transpose(2, 1, 0).reshape(-1), c["lambda_um"], LW_SURFACE_TEMP, dl)
        probs.append(P); srcs.append(O.EmissionSource(vw, frac)); fl.append(f)
    cdf, flux = broadband.emitted_flux_cdf(fl)
    rng = O.mt_rng([seed, proc, 0])
    t1 = time.time()
    counts = O.frequency_distribution(rng, cdf, total)
    rows, counters = [], None
    for P, src, cnt in zip(probs, srcs, counts):
        left = int(cnt)
        while left > 0:
            k = min(ppb, left)
            r = O.compute_radiative_transfer(P, src, rng, k)
            rows.append((k, np.array([r["meanFluxUp"], r["meanFluxDown"], r["meanFluxAbsorbed"]], np.float64),
                         np.concatenate([r["fluxUp"], r["fluxDown"], r["fluxAbsorbed"]]).astype(np.float64),
                         np.asarray(r["absorbedProfile"], np.float64)))
            counters = r["counters"] if counters is None else {q: counters[q] + v for q, v in r["counters"].items()}
            left -= k
    return rows, flux, counters, time.time() - t1, t1 - t0


def oracle_lw_run(total_per_proc, ppb, procs, lambdas=LW_LAMBDAS, n=20, seed=10):
    """Config 4 on the host cores, oracle in MT mode: every process runs the whole spectrum with its own MT stream.
    -> (dict quantity -> list of (n, values) per batch, spectrally integrated emitted flux, event counters, seconds of
    the busiest process in the photon loop)."""
    from oracle import oracle as O
    O.build()
    jobs = [(tuple(lambdas), n, seed, p + 1, int(total_per_proc), int(ppb)) for p in range(procs)]
    if procs == 1:
        parts = [_lw_worker(jobs[0])]
    else:
        with mp.get_context("spawn").Pool(procs) as pool:
            parts = pool.map(_lw_worker, jobs)
    rows = [r for part in parts for r in part[0]]
    counters = {q: sum(part[2][q] for part in parts) for q in parts[0][2]}
    out = {"means": [(k, a) for k, a, _, _ in rows], "columns": [(k, c) for k, _, c, _ in rows],
           "profile": [(k, p) for k, _, _, p in rows]}
    return out, parts[0][1], counters, max(part[3] for part in parts)


def lw_mean_err(batches, flux):
    from oracle import oracle as O
    return {q: O.batch_statistics(batches[q], solar_flux=flux) for q in QUANTITIES}
