"""Photon pool (trace_kernel<..., POOL>) against the same kernel without it: moment arrays bit for bit, and the rates.
usage: python scripts/pool_check.py [case] [photons per batch] [batches]   (case: landsat | radar | small)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from tests import cases  # noqa: E402


def run(case, pool, ppb, nb, mu0=0.5, phi0=30.0, reps=2, env=None):
    os.environ["MCBRAT_POOL"] = str(pool)  # read when the context is created
    for k, v in (env or {}).items():
        os.environ[k] = str(v)
    import mcbrat3d_amd as M
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=10001)
    integ.setTuning(eventThreshold=int(os.environ.get("THR", "20")))
    photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 12)
    best = 0.0
    for r in range(reps):
        integ.resetMoments()
        t = time.time()
        n = integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(4242), photons, ppb, nb)
        integ.synchronize()
        best = max(best, n / (integ.lastTraceMs() * 1e-3))
    mom = integ.moments().copy()
    bad = integ.badPhotons()
    integ.finalize()
    return mom, best, bad


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "landsat"
    ppb = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    case = {"landsat": lambda: cases.landsat_like(), "radar": lambda: cases.radar_like(),
            "small": lambda: cases.landsat_like(n=32, nz=24, n_entries=6)}[name]()
    m0, r0, b0 = run(case, 0, ppb, nb)
    print("%s pool=0: %.4g photons/s (kernel), bad %d" % (name, r0, b0), flush=True)
    for env in ([{}] if len(sys.argv) <= 4 else [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[4:]]):
        m1, r1, b1 = run(case, 1, ppb, nb, env=env)
        same = np.array_equal(m0, m1)
        print("%s pool=1 %s: %.4g photons/s (kernel), bad %d, moments bitwise equal: %s (photons %d vs %d)" % (
            name, env, r1, b1, same, m0[0], m1[0]), flush=True)
        if not same:
            d = np.flatnonzero(m0 != m1)
            print("   %d of %d elements differ; first at %d: %r vs %r; means %r vs %r" % (d.size, m0.size, d[0], m0[d[0]], m1[d[0]], m0[8:11] / m0[0], m1[8:11] / m1[0]))


if __name__ == "__main__":
    main()
