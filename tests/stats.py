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
