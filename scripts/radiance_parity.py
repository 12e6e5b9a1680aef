"""Radiance parity at statistics: the GPU (Philox, layer-skipping rays, ray buffer) against the CPU oracle in MT
mode (the reference's generator and draw order), one oracle process per host core; per direction the domain-mean
radiance and the per-pixel z-scores.  usage: radiance_parity.py [photons per core] [gpu photons]"""
import json
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from tests import cases  # noqa: E402

MUS, PHIS = [1.0, 0.6, 0.3], [0.0, 120.0, 300.0]
MU0, PHI0 = 0.5, 30.0
BATCH = 20000


def make_case():
    return cases.landsat_like(n=32, nz=24, n_entries=6, albedo=0.2)


def worker(args):
    n, proc = args
    from oracle import oracle as O
    case = make_case()
    P = cases.oracle_problem(case, nsteps=9001)
    I = cases.oracle_intensity(case, MUS, PHIS, n_angles=9001, use_russian_roulette=True, zeta_min=0.3)
    rng = O.mt_rng([10, proc, 0])
    out, done = [], 0
    while done < n:
        nb = min(BATCH, n - done)
        r = O.compute_radiative_transfer_intensity(P, O.solar_source(MU0, PHI0), rng, nb, I)
        out.append((nb, np.asarray(r["intensity"], np.float64).reshape(-1)))
        done += nb
    return out


if __name__ == "__main__":
    per_core = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    n_gpu = int(sys.argv[2]) if len(sys.argv) > 2 else 20000000
    import mcbrat3d_amd as M
    from mcbrat3d_amd import driver
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    from oracle import oracle as O
    O.build()
    case = make_case()
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=9001, minForwardTableSize=9001, intensityMus=MUS, intensityPhis=PHIS,
                            computeIntensity=True, useRussianRouletteForIntensity=True, zetaMin=0.3)
    photons = M.new_PhotonStream(MU0, PHI0, numberOfPhotons=10 ** 12)
    integ.resetMoments()
    t = time.time()
    integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(10), photons, 200000, n_gpu // 200000)
    st = driver.statistics(driver.unpack_moments(integ.moments(), dom.numX, dom.numY, dom.numZ, len(MUS)))
    print("GPU: %d photons in %.2f s" % (st["totalPhotons"], time.time() - t), flush=True)
    cores = min(16, len(os.sched_getaffinity(0)))
    t = time.time()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(worker, [(per_core, p + 1) for p in range(cores)])
    batches = [b for o in res for b in o]
    mean, err = O.batch_statistics(batches)
    print("CPU oracle (MT): %d photons on %d cores in %.1f s" % (sum(b[0] for b in batches), cores, time.time() - t), flush=True)
    g = np.concatenate([st["intensity"][:, :, d].T.reshape(-1) for d in range(len(MUS))])
    ge = np.concatenate([st["intensity_StdErr"][:, :, d].T.reshape(-1) for d in range(len(MUS))])
    z = (g - mean) / np.sqrt(ge ** 2 + err ** 2 + 1e-30)
    npix = dom.numX * dom.numY
    rec = {"gpu_photons": int(st["totalPhotons"]), "cpu_photons": int(sum(b[0] for b in batches)), "directions": list(zip(MUS, PHIS))}
    for d in range(len(MUS)):
        sl = slice(d * npix, (d + 1) * npix)
        gm, cm = float(np.mean(g[sl])), float(np.mean(mean[sl]))
        # error of a domain mean from the batch means of the domain mean
        cb = [(n, np.array([np.mean(v[sl])])) for n, v in batches]
        _, ce = O.batch_statistics(cb)
        print("direction mu=%.1f: mean radiance GPU %.6f CPU %.6f (CPU stderr %.1e, z %.2f); pixels: z mean %.3f std %.3f max |z| %.2f" % (
            MUS[d], gm, cm, ce[0], (gm - cm) / ce[0], np.mean(z[sl]), np.std(z[sl]), np.max(np.abs(z[sl]))), flush=True)
        rec["mu_%g" % MUS[d]] = dict(gpu_mean=gm, cpu_mean=cm, cpu_stderr=float(ce[0]), z_mean=float((gm - cm) / ce[0]),
                                     pixel_z_mean=float(np.mean(z[sl])), pixel_z_std=float(np.std(z[sl])), pixel_z_max=float(np.max(np.abs(z[sl]))))
    print(json.dumps(rec))
