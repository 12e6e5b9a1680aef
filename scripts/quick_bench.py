"""Development timing loop (not the judged bench): step cloud / landsat-like, sweeps of tuning knobs."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from tests import cases  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="step")
    ap.add_argument("--ppb", type=int, default=100000)
    ap.add_argument("--batches", type=int, default=100)
    ap.add_argument("--thr", type=int, nargs="*", default=[40])
    ap.add_argument("--bpc", type=int, nargs="*", default=[0])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--ssa", type=float, default=0.99)
    ap.add_argument("--counters", action="store_true")
    ap.add_argument("--inflight", type=int, nargs="*", default=[-1])
    ap.add_argument("--lthr", type=int, nargs="*", default=[0])
    ap.add_argument("--sthr", type=int, nargs="*", default=[0])
    ap.add_argument("--priv", type=int, nargs="*", default=[-1])
    ap.add_argument("--brick", type=int, nargs="*", default=[-1])
    ap.add_argument("--block", type=int, nargs="*", default=[-1])
    ap.add_argument("--skip", type=int, nargs="*", default=[-1])
    ap.add_argument("--bw", type=int, nargs="*", default=[-1], help="block walk (LDS-resident grids): -1 default, 0 off, 1 on")
    ap.add_argument("--n", type=int, default=128, help="landsat / radar: columns per side")
    ap.add_argument("--nz", type=int, default=64)
    ap.add_argument("--opt", nargs="*", default=[], help="scheduling options by name, e.g. jumpThreshold=8,16 (comma: sweep)")
    ap.add_argument("--haze", type=float, default=100.0, help="case hazy: the Rayleigh component times this (optical depth 0.023 x haze)")
    a = ap.parse_args()
    import mcbrat3d_amd as M
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    if a.case == "step":
        case = cases.step_cloud(a.ssa)
    elif a.case == "plane":
        case = cases.plane_parallel(ssa=a.ssa)
    elif a.case == "radar":
        case = cases.radar_like(n=a.n, nz=a.nz)
    elif a.case == "overcast":  # a stratus deck: every column cloudy over the same layers, clear air (Rayleigh) above and below
        case = cases.landsat_like(n=a.n, nz=a.nz, ssa_cloud=a.ssa)
        e = case["components"][0]["ext"]
        col = e.sum(axis=2, keepdims=True) / 12.0
        e[:] = 0.0
        e[:, :, 8:20] = col
        case["components"][0]["ssa"] = np.where(e > 0, a.ssa, 0.0)
    elif a.case == "hazy":  # the cloud field in a haze one can collide in (the second component a hundred times the Rayleigh one)
        case = cases.landsat_like(n=a.n, nz=a.nz, ssa_cloud=a.ssa)
        case["components"][1]["ext"] = a.haze * case["components"][1]["ext"]
    else:
        case = cases.landsat_like(n=a.n, nz=a.nz, ssa_cloud=a.ssa)
    mu0, phi0 = (1.0, 0.0) if a.case in ("step", "plane") else (0.5, 30.0)
    t0 = time.time()
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=10001)
    photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 12)
    print("setup %.2fs" % (time.time() - t0), flush=True)
    import itertools
    opt_names = [o.split("=")[0] for o in a.opt]
    opt_values = [[int(v) for v in o.split("=")[1].split(",")] for o in a.opt]
    for bpc, priv, block, thr, lthr, sthr, brick, skip, bw in itertools.product(a.bpc, a.priv, a.block, a.thr, a.lthr, a.sthr, a.brick, a.skip, a.bw):
      for inflight in a.inflight:
        for combo in itertools.product(*opt_values):
            opts = dict(zip(opt_names, combo))
            integ.setOption(**opts)
            integ.setTuning(blocksPerCU=bpc, eventThreshold=thr, privateTallies=priv, blockSize=block, launchThreshold=lthr, surfaceThreshold=sthr, brickLayout=brick, maxBatchesInFlight=inflight, layerSkip=skip, blockWalk=bw)
            rates = []
            for r in range(a.reps):
                rng = new_RandomNumberSequence(1234 + r)
                integ.resetMoments()
                t = time.time()
                n = integ.computeRadiativeTransfer(dom, rng, photons, a.ppb, a.batches)
                dt = time.time() - t
                rates.append((n / dt, n / (integ.lastTraceMs() * 1e-3)))
            if a.counters:
                integ.enableCounters(True)
                integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(99), photons, a.ppb, a.batches)
                c = integ.counters(); integ.enableCounters(False)
                print('   per photon: legs %.2f crossings %.2f collisions %.2f' % (c['legs'] / n, c['crossings'] / n, c['collisions'] / n))
                print('   walk iters %.4g lanes/iter %.1f | event phases %.4g lanes/phase %.1f | launch phases %.4g surf phases %.4g | walk iters per event phase %.2f' % (c['walkIterations'], c['walkLanes']/max(1,c['walkIterations']), c['eventPhases'], c['eventLanes']/max(1,c['eventPhases']), c['launchPhases'], c['surfacePhases'], c['walkIterations']/max(1,c['eventPhases'])))
            res = integ.reportResults()
            if thr == 0:
                print("   (event threshold chosen by the trial launches: %d)" % integ.eventThreshold())
            print("opts=%s " % opts, end="")
            print("case=%s n=%d bw=%d skip=%d bpc=%d priv=%d block=%d thr=%d lthr=%d sthr=%d brick=%d inflight=%d ppb=%d nb=%d  wall %.3g ph/s  kernel %.3g ph/s  (means %.5f %.5f %.5f)" % (
                a.case, a.n, bw, skip, bpc, priv, block, thr, lthr, sthr, brick, inflight, a.ppb, a.batches, max(r[0] for r in rates), max(r[1] for r in rates),
                res["meanFluxUp"], res["meanFluxDown"], res["meanFluxAbsorbed"]), flush=True)


if __name__ == "__main__":
    main()
