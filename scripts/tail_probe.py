"""What a launch's tail consists of: walk iterations per photon (library built with -DMCBRAT_FATE_STEPS: the fates' nEvents
field counts walk iterations) on the 128x128x64 cloud field, and the tracing-kernel time of small launches.
usage: MCBRAT_LIB=ab/libmcbrat_steps.so python scripts/tail_probe.py [photons]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from tests import cases  # noqa: E402
import mcbrat3d_amd as M  # noqa: E402
from mcbrat3d_amd.integrator import new_RandomNumberSequence  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4000000
case = cases.landsat_like()
dom = cases.product_domain(case)
for skip in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2)):
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=10001)
    integ.setTuning(eventThreshold=20, layerSkip=skip)
    photons = M.new_PhotonStream(0.5, 30.0, numberOfPhotons=10 ** 12)
    f = integ.traceFates(dom, new_RandomNumberSequence(4242), photons, n)
    st = f["nEvents"].astype(np.int64)
    order = np.argsort(st)[::-1][:8]
    print("layerSkip %d: walk iterations per photon: mean %.1f p99 %d p99.9 %d p99.99 %d max %d; sum of the top 100: %d; worst: %s" % (
        skip, st.mean(), np.quantile(st, 0.99), np.quantile(st, 0.999), np.quantile(st, 0.9999), st.max(), np.sort(st)[-100:].sum(),
        [(int(i), int(st[i]), int(f["fate"][i]), int(f["nScatter"][i])) for i in order]), flush=True)
    for m in (250000, 1000000, 4000000, 16000000):
        integ.resetMoments()
        integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(7), photons, m // 10, 10)
        integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(8), photons, m // 10, 10)
        print("   %9d photons per launch: kernel %.3f ms" % (m, integ.lastTraceMs()), flush=True)
    integ.finalize()
