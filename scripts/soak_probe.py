"""Development probe: diagnostics of seeds of tests/test_gpu_layer_skip.py::test_random_domains_against_face_by_face_walk
whose per-photon identity with the face-by-face walk falls below the test's 97 %: identity of the SHORT histories and the
flux differences tell a chaotic medium (long histories flip on a rounding difference) from a wrong walk."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcbrat3d_amd as M
from tests import cases
from tests.test_gpu_layer_skip import _integ, _same

for seed in [int(x) for x in sys.argv[1:]]:
    rng = np.random.default_rng(1000 + seed)
    nx, ny, nz = int(rng.integers(1, 9)), int(rng.integers(1, 7)), int(rng.integers(2, 14))
    def edges(n, stretched):
        d = rng.uniform(0.02, 0.06) * (np.cumprod(rng.uniform(0.85, 1.2, n)) if stretched else np.ones(n))
        return np.concatenate([[0.0], np.cumsum(d)]) + (rng.uniform(-1.0, 1.0) if stretched else 0.0)
    xe, ye, ze = edges(nx, rng.random() < 0.4), edges(ny, rng.random() < 0.4), edges(nz, rng.random() < 0.5)
    ze -= ze[0]
    ext = rng.uniform(0.0, 25.0, (nx, ny, nz))
    uniform = rng.random(nz) < rng.choice([0.0, 0.5, 0.8, 1.0])
    for k in np.nonzero(uniform)[0]:
        ext[:, :, k] = rng.choice([0.0, rng.uniform(0.01, 8.0)])
    ssa0 = rng.uniform(0.6, 1.0)
    comps = [dict(ext=ext, ssa=np.where(ext > 0, ssa0, 0.0), pfIndex=np.ones(ext.shape, np.int32),
                  legendre=[cases.hg_legendre(rng.uniform(0.0, 0.9), 24)])]
    if rng.random() < 0.5:
        comps.append(dict(ext=rng.uniform(0.0, 0.3, nz), ssa=np.ones(nz), pfIndex=np.ones(nz, np.int32),
                          legendre=[np.array([0.0, 0.1], np.float32)]))
    case = dict(name="random%d" % seed, xe=xe, ye=ye, ze=ze, albedo=float(rng.choice([0.0, 0.3, 0.8])), components=comps)
    mu0, phi0 = float(rng.choice([1.0, rng.uniform(0.05, 1.0)])), float(rng.uniform(0.0, 360.0))
    n = 20000
    fates, means = {}, {}
    for skip in (0, 1):
        dom, integ, photons, r = _integ(M, case, mu0, phi0, skip, rr=bool(rng.integers(0, 2)) if skip == 0 else rr_used)
        rr_used = integ.useRussianRoulette
        fates[skip] = integ.traceFates(dom, r, photons, n)
        integ.resetMoments()
        integ.computeRadiativeTransfer(dom, r, photons, 200000)
        res = integ.reportResults()
        means[skip] = np.array([res["meanFluxUp"], res["meanFluxDown"], res["meanFluxAbsorbed"]])
        integ.finalize()
    same = _same(fates[1], fates[0])
    ns = fates[0]["nScatter"]
    print("seed %d: %dx%dx%d ssa %.2f albedo %.1f mu0 %.2f roulette %s uniform layers %d/%d | identical %.4f; by scattering order <=2: %.4f, 3-10: %.4f, >10: %.4f (mean order %.1f) | flux differences at 2e5 photons %s" % (
        seed, nx, ny, nz, ssa0, case["albedo"], mu0, rr_used, uniform.sum(), nz, same.mean(), same[ns <= 2].mean(), same[(ns > 2) & (ns <= 10)].mean() if np.any((ns > 2) & (ns <= 10)) else float("nan"),
        same[ns > 10].mean() if np.any(ns > 10) else float("nan"), ns.mean(), np.round(means[1] - means[0], 5)), flush=True)
