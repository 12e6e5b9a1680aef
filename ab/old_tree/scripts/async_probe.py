"""Development probe: many per-batch calls in synchronous / asynchronous mode (wall time per call)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases  # noqa: E402
import mcbrat3d_amd as M  # noqa: E402
from mcbrat3d_amd.integrator import new_RandomNumberSequence  # noqa: E402

case_name = sys.argv[1] if len(sys.argv) > 1 else "step"
ppb = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 40
reset_each = len(sys.argv) > 5 and "reset" in sys.argv[5]
bind = len(sys.argv) > 5 and "bind" in sys.argv[5]
if bind:
    import torch
    torch.cuda.init()
    torch.zeros(1, device="cuda")
case, mu0, phi0 = (cases.step_cloud(0.99), 1.0, 0.0) if case_name == "step" else (cases.landsat_like(), 0.5, 30.0)
dom = cases.product_domain(case)
photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 12)
for mode in ("sync", "async"):
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=10001)
    integ.setTuning(eventThreshold=16 if case_name == "step" else 32)
    integ.setAsync(mode == "async")
    rng = new_RandomNumberSequence(7)
    if bind:
        mom = torch.zeros(8 + 2 * integ.momentsLength(), dtype=torch.float64, device="cuda")
        integ.bindMoments(mom.data_ptr())
    integ.resetMoments()
    for k in range(4):
        integ.computeRadiativeTransfer(dom, rng, photons, ppb, nb)
    integ.synchronize()
    t0 = time.time()
    for k in range(calls):
        if reset_each:
            integ.resetMoments()
        integ.computeRadiativeTransfer(dom, rng, photons, ppb, nb)
    t1 = time.time()
    integ.synchronize()
    t2 = time.time()
    print("%s %s: %d calls of %d x %d photons: enqueue %.2f ms, total %.2f ms -> %.3g photons/s; kernel time sum %.2f ms" % (
        case_name, mode, calls, nb, ppb, (t1 - t0) * 1e3, (t2 - t0) * 1e3, calls * nb * ppb / (t2 - t0), integ.lastTraceMs()), flush=True)
    integ.finalize()
