"""Development check: radiance with the long rays put aside (default) against every ray finished in its event phase
(MCBRAT_RAY_DEFER=0): the moment arrays must be bitwise equal; rates at 2e7 photons per launch."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from tests import cases
    import mcbrat3d_amd as M
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    which, ndir, rr, nb = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    case = cases.landsat_like() if which == "landsat128" else (cases.landsat_like(n=32, nz=24, n_entries=6, albedo=0.3) if which == "landsat32" else cases.step_cloud(0.99))
    mu0, phi0 = (1.0, 0.0) if which == "step" else (0.5, 30.0)
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    mus = np.linspace(1.0, 0.3, ndir); phis = np.linspace(0.0, 300.0, ndir)
    integ.specifyParameters(minInverseTableSize=9001, intensityMus=mus, intensityPhis=phis, computeIntensity=True, useRussianRouletteForIntensity=bool(rr))
    integ.setTuning(eventThreshold=32)
    photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 12)
    integ.resetMoments()
    n = integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(5), photons, 1000000 if nb > 1 else 50000, nb)
    np.save(sys.argv[6], integ.moments())
    print("%.1f ms" % integ.lastTraceMs())
    sys.exit(0)
for which, ndir, rr, nb in (("landsat32", 3, 1, 1), ("landsat32", 3, 0, 1), ("step", 2, 1, 1), ("landsat128", 1, 1, 20), ("landsat128", 4, 1, 20), ("landsat128", 4, 0, 10)):
    out = {}
    for defer in (0, 1):
        env = dict(os.environ, MCBRAT_RAY_DEFER=str(defer))
        f = "/tmp/raydefer_%d.npy" % defer
        r = subprocess.run([sys.executable, __file__, "child", which, str(ndir), str(rr), str(nb), f], env=env, capture_output=True, text=True, timeout=600)
        out[defer] = (np.load(f), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
    print("%-10s ndir=%d roulette=%d: in phase %s, put aside %s, bitwise equal: %s" % (which, ndir, rr, out[0][1], out[1][1], np.array_equal(out[0][0], out[1][0])), flush=True)
