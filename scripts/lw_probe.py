"""Config 4 (thermal, 20x20x20): event counters and -- with a library built -DMCBRAT_STAMPS -- wave cycles per section of
the loop for ONE wavelength's photons; kernel time of the production instantiation beside it.
usage: [MCBRAT_LIB=ab/libmcbrat_stamps.so] python scripts/lw_probe.py [photons per batch] [batches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases, stats as tstats  # noqa: E402
import torch  # noqa: E402,F401  (before the product initialises the GPU: two HIP runtimes in one process)
import mcbrat3d_amd as M  # noqa: E402
from mcbrat3d_amd import broadband  # noqa: E402
from mcbrat3d_amd.integrator import new_RandomNumberSequence  # noqa: E402

ppb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 6
doms = [cases.product_domain(c) for c in tstats.lw_cases()][:1]
run = broadband.SpectralRun(M, doms, minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=True)
it = run.first
it.setTuning(eventThreshold=int(os.environ.get("THR", "0")), launchThreshold=int(os.environ.get("LTHR", "0")), surfaceThreshold=int(os.environ.get("STHR", "0")))
run.prepare_thermal(tstats.LW_SURFACE_TEMP)
rng = new_RandomNumberSequence(10)
ps = run.streams[0]
for counters in (False, True):
    it.enableCounters(counters)
    for rep in range(3):
        ps.currentPhoton = 1
        it.resetMoments()
        it.computeRadiativeTransfer(doms[0], rng, ps, ppb, nb)
        it.synchronize()
        print("counters %s: %d photons, tracing kernel %.3f ms -> %.4g photons/s" % (counters, ppb * nb, it.lastTraceMs(), ppb * nb / it.lastTraceMs() * 1e3), flush=True)
    if counters:
        c = it.counters()
        n = float(ppb * nb)
        print({k: round(v / n, 4) for k, v in c.items()})
        print("lanes per walk iteration %.1f, per event phase %.1f; wave-level walk iterations per photon %.3f, event phases %.3f, launch phases %.3f, exit phases %.3f" % (
            c["walkLanes"] / max(1, c["walkIterations"]), c["eventLanes"] / max(1, c["eventPhases"]), c["walkIterations"] / n, c["eventPhases"] / n,
            c["launchPhases"] / n, c["surfacePhases"] / n))
run.finalize()
