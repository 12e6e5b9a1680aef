import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import cases
import mcbrat3d_amd as M
from mcbrat3d_amd.integrator import new_RandomNumberSequence
case = cases.landsat_like(n=48, nz=24, n_entries=6)
dom = cases.product_domain(case)
for wd in (1 << 20, 20000):
    os.environ["MCBRAT_WATCHDOG"] = str(wd)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=9001)
    integ.setTuning(eventThreshold=20)
    photons = M.new_PhotonStream(0.5, 30.0, numberOfPhotons=10 ** 9)
    for n in (2000, 200000):
        integ.resetMoments()
        t = time.time()
        integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(90210), photons, n)
        r = integ.reportResults()
        print("watchdog %d: %d photons: kernel %.2f ms, bad %d, means %.5f %.5f %.5f sum %.5f" % (
            wd, n, integ.lastTraceMs(), integ.badPhotons(), r["meanFluxUp"], r["meanFluxDown"], r["meanFluxAbsorbed"],
            r["meanFluxUp"] + r["meanFluxDown"] + r["meanFluxAbsorbed"]), flush=True)
    f = integ.traceFates(dom, new_RandomNumberSequence(90210), photons, 200000)
    print("   fates:", np.bincount(np.maximum(f["fate"], 0), minlength=4), "bad", integ.badPhotons(), "dropped ids", np.flatnonzero(f["fate"] == 3)[:10], flush=True)
    integ.finalize()
