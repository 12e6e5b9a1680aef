"""The ORACLE's reference-faithful mode (MT19937 streams, the reference's draw order and rejection loop) against transport
theory: the slabs of scripts/analytic_probe.py, ANALYTIC_ORACLE_PHOTONS photons each (default 4e7) over all host cores.
CPU only; test infrastructure, like everything that touches oracle/."""
import os, sys
from concurrent.futures import ProcessPoolExecutor
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases
from tests.test_analytic import (isotropic_slab, doubling_slab, sampled_moments, slab, hg_slab, tabulated_slab, SCATTERING_SLABS,
                                 HG_SLABS, HG_STREAMS)

PER = 10000      # photons per batch: the reference's tallies are single precision (Integrators/monteCarloRadiativeTransfer.f95:102), and
                 # the oracle's with them -- 10^6 weights of a non-conservative medium added into one column's float lose 10^-3
CHUNK = 50       # batches per task


def one(args):
    kind, params, mu0, phi0, seed = args
    from oracle import oracle as O
    if kind == "iso":
        case = slab(params[0], params[1], nz=16)
    elif kind == "hg":
        case = hg_slab(*params)[0]
    else:
        case = tabulated_slab(*params)[0]
    P = cases.oracle_problem(case, nsteps=9001)
    out = []
    for j in range(CHUNK):
        r = O.compute_radiative_transfer(P, O.solar_source(mu0, phi0), O.mt_rng(seed * CHUNK + j), PER)
        out.append((float(r["meanFluxUp"]), float(r["meanFluxDown"])))
    return out


def run(pool, kind, params, mu0, phi0, label, up, down, total):
    nt = max(2, total // (PER * CHUNK))
    res = np.array([b for t in pool.map(one, [(kind, params, mu0, phi0, 1000 + k) for k in range(nt)]) for b in t])
    nb = len(res)
    mean, err = res.mean(axis=0), res.std(axis=0, ddof=1) / np.sqrt(nb)
    print("%s n %.0e (MT mode, batches of %d): up %.6f +- %.6f theory %.6f z %.2f | down %.6f +- %.6f theory %.6f z %.2f" % (
        label, nb * PER, PER, mean[0], err[0], up, (mean[0] - up) / err[0], mean[1], err[1], down, (mean[1] - down) / err[1]), flush=True)


if __name__ == "__main__":
    total = int(float(os.environ.get("ANALYTIC_ORACLE_PHOTONS", "4e7")))
    with ProcessPoolExecutor(os.cpu_count()) as pool:
        for b, omega, mu0 in SCATTERING_SLABS:
            up, down, direct = isotropic_slab(b, omega, mu0, cells=3000)
            run(pool, "iso", (b, omega), mu0, 75.0, "isotropic b %.1f omega %.2f mu0 %.3f" % (b, omega, mu0), up, down + direct, total)
        for b, omega, g, nleg, node in HG_SLABS:
            chi = hg_slab(b, omega, g, nleg)[1]
            mu0, up, down = doubling_slab(b, omega, sampled_moments(chi, table=9001), node, streams=HG_STREAMS)
            run(pool, "hg", (b, omega, g, nleg), mu0, 20.0, "HG g %.2f (%d terms) b %.1f omega %.2f mu0 %.4f" % (g, nleg, b, omega, mu0), up, down, total)
        nodes = tabulated_slab(2.0, 0.9)[1]
        mu0, up, down = doubling_slab(2.0, 0.9, sampled_moments(None, table=9001, nodes=nodes), 64, streams=HG_STREAMS)
        run(pool, "tab", (2.0, 0.9), mu0, 20.0, "angle / value table b 2.0 omega 0.90 mu0 %.4f" % mu0, up, down, total)
