#!/usr/bin/env python3
"""bench.py -- photons/sec of the photon-tracing hot path on N MI355X of one node.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one complete job of the workload on every rank: the I3RC step cloud
(BASELINE.json configs[1]: 32x1x32, tau 2|18, omega0 0.99, HG g 0.85, mu0 1) traced with
1e7 photons per GPU as 100 batches of 1e5 (computeRadiativeTransfer + reportResults +
batch moments on the device), followed by the all-reduce of the moment arrays over RCCL
(the reference's sumAcrossProcesses).  Weak scaling: every rank traces its own 1e7
photons per step (disjoint photon-id ranges of one Philox key).  Inputs (grids, tables)
are resident in HBM before the timed region; photons are generated on the GPU.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- algorithmic bytes of the tracing kernel / its HIP-event duration vs HBM peak
  cpu_baseline -- the CPU oracle (a port of the reference loop, pinned to the reference's
                  own outputs) timed on this box's host cores on a bounded photon sample
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tests import cases  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
WORKLOADS = {
    # cpu: photons per host core of the cpu_baseline sample (about 20 s of oracle time), parity: photons of the GPU parity run
    "i3rcStepCloud": dict(make=lambda: cases.step_cloud(ssa=0.99), mu0=1.0, phi0=0.0, ppb=100000, batches=100,
                          cpu=8000000, parity=400000000),
    "landsatLike128": dict(make=lambda: cases.landsat_like(), mu0=0.5, phi0=30.0, ppb=1000000, batches=100,
                           cpu=2000000, parity=100000000),
    # config 5: optically thick, omega0 = 0.9, roulette-heavy; not a headline line, kept for its parity record
    "radarLike128": dict(make=lambda: cases.radar_like(), mu0=0.5, phi0=30.0, ppb=1000000, batches=100,
                         cpu=4000000, parity=100000000),
}


def algorithmic_bytes_per_photon(c, n, nc):
    """B = 4 N_cross + (4 nc + 14) N_coll + 16 N_abs + 8 N_exit (per photon)."""
    return (4.0 * c["crossings"] + (4.0 * nc + 14.0) * c["collisions"] + 16.0 * c["absorbEvents"] +
            8.0 * (c["topExits"] + c["surfaceHits"])) / n


CPU_BATCH = 100000  # the reference tallies in float32: batches must stay small


def _cpu_worker(args):
    """cpu_baseline leg: the ORACLE, one process per core, one MT stream per process seeded
    (/iseed, proc, 0/) and carried across batches exactly as the reference driver does
    (monteCarloDriver.f95:901), batches of 1e5 photons."""
    name, n, proc = args
    from oracle import oracle as O
    w = WORKLOADS[name]
    P = cases.oracle_problem(w["make"]())
    rng = O.mt_rng([10, proc, 0])
    src = O.solar_source(w["mu0"], w["phi0"])
    t = time.time()
    means, cols, counters, done = [], [], None, 0
    while done < n:
        nb = min(CPU_BATCH, n - done)
        res = O.compute_radiative_transfer(P, src, rng, nb)
        means.append((nb, np.array([res["meanFluxUp"], res["meanFluxDown"], res["meanFluxAbsorbed"]], np.float64)))
        cols.append((nb, np.concatenate([res["fluxUp"], res["fluxDown"], res["fluxAbsorbed"]]).astype(np.float64)))
        counters = res["counters"] if counters is None else {k: counters[k] + v for k, v in res["counters"].items()}
        done += nb
    return time.time() - t, n, means, cols, counters


def cpu_baseline(name, photons_per_core, max_cores):
    from oracle import oracle as O
    O.build()
    cores = max(1, min(max_cores, len(os.sched_getaffinity(0))))
    t = time.time()
    with mp.get_context("spawn").Pool(cores) as pool:
        out = pool.map(_cpu_worker, [(name, photons_per_core, p + 1) for p in range(cores)])
    wall = time.time() - t
    total = sum(o[1] for o in out)
    busiest = max(o[0] for o in out)
    batches = [b for o in out for b in o[2]]
    cols = [b for o in out for b in o[3]]
    counters = {k: sum(o[4][k] for o in out) for k in out[0][4]}
    return dict(value=total / busiest, unit="photons/s", cores=cores, kind="port",
                sample="%d photons/core x %d cores of the same workload in batches of %d, oracle in MT mode "
                       "(%.1f s wall)" % (photons_per_core, cores, CPU_BATCH, wall)), batches, cols, counters, total


def secondary_workload(M, new_rng, name="landsatLike128", steps=3):
    """Untimed extra (not the contract's metric): BASELINE.json's target is quoted on a 128x128x64 domain, so the
    default single-GPU run also reports that workload's rate: `steps` synchronous steps of 1e8 photons after one
    warm-up step (which also lets the library choose its event threshold)."""
    w = WORKLOADS[name]
    dom = cases.product_domain(w["make"]())
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=10001, useRayTracing=True, useRussianRoulette=True)
    photons = M.new_PhotonStream(w["mu0"], w["phi0"], numberOfPhotons=10 ** 15)
    rng = new_rng(10)
    integ.resetMoments()
    integ.computeRadiativeTransfer(dom, rng, photons, w["ppb"], w["batches"])
    integ.synchronize()
    t0, kms = time.perf_counter(), 0.0
    for _ in range(steps):
        integ.resetMoments()
        integ.computeRadiativeTransfer(dom, rng, photons, w["ppb"], w["batches"])
        kms += integ.lastTraceMs()
    integ.synchronize()
    dt = time.perf_counter() - t0
    integ.enableCounters(True)
    integ.computeRadiativeTransfer(dom, rng, photons, w["ppb"], 10)
    cnt = integ.counters()
    integ.enableCounters(False)
    per_step = w["ppb"] * w["batches"]
    bpp = algorithmic_bytes_per_photon(cnt, w["ppb"] * 10, len(dom.components))
    achieved = bpp * per_step / (kms / steps * 1e-3) / 1e9
    res = {"workload": "%s %dx%dx%d, %d photons/step" % (name, dom.numX, dom.numY, dom.numZ, per_step),
           "value": per_step * steps / dt, "unit": "photons/s", "steps": steps, "ms_per_step": 1e3 * dt / steps,
           "kernel_ms_per_launch": kms / steps, "algorithmic_bytes_per_photon": bpp,
           "roofline_frac": achieved / HBM_PEAK_GBS, "event_threshold": integ.eventThreshold()}
    integ.finalize()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="i3rcStepCloud", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-photons-per-core", type=int, default=0, help="0 = the workload's default")
    ap.add_argument("--cpu-cores", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the untimed 128x128x64 extra of the default run")
    ap.add_argument("--parity-photons", type=int, default=0, help="0 = the workload's default")
    ap.add_argument("--event-threshold", type=int, default=0,
                    help="0 = let the library time trial launches (default); >0 fixes it (profiling runs)")
    ap.add_argument("--pipeline", action="store_true",
                    help="let consecutive steps overlap on the GPU (asynchronous mode of the library: hides each "
                         "launch's drain; per-kernel durations then include time spent waiting for compute units, so "
                         "the default run, whose kernel durations rocprofv3 must reproduce, does not use it)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        a.no_cpu_baseline = True  # the CPU baseline (and the parity check against it) belongs to the N = 1 line
    import torch
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):  # BENCH_FORCE_DIST: exercise the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        # RCCL prints a version banner on stdout when its communicator comes up; the contract is ONE JSON line there,
        # so file descriptor 1 points at stderr until the first collective is through
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            warm = torch.zeros(1, device=torch.device("cuda", local_rank))
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the photon-tracing path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import mcbrat3d_amd as M
    from mcbrat3d_amd import driver
    from mcbrat3d_amd.integrator import new_RandomNumberSequence

    w = WORKLOADS[a.workload]
    a.cpu_photons_per_core = a.cpu_photons_per_core or w["cpu"]
    a.parity_photons = a.parity_photons or w["parity"]
    case = w["make"]()
    dom = cases.product_domain(case)
    nx, ny, nz = dom.numX, dom.numY, dom.numZ
    nc = len(dom.components)
    integ = M.new_Integrator(dom, device=local_rank)
    integ.specifyParameters(minInverseTableSize=10001, useRayTracing=True, useRussianRoulette=True)
    integ.setTuning(eventThreshold=a.event_threshold)
    photons = M.new_PhotonStream(w["mu0"], w["phi0"], numberOfPhotons=10 ** 15)
    ppb, nb = w["ppb"], w["batches"]
    per_step = ppb * nb
    moments = torch.zeros(8 + 2 * integ.momentsLength(), dtype=torch.float64, device=dev)
    integ.bindMoments(moments.data_ptr())
    rng = new_RandomNumberSequence(10)
    torch_stream = torch.cuda.current_stream(dev).cuda_stream

    def step(i):
        # rank r traces photon ids [ (i*world + r) * per_step, +per_step ): disjoint over ranks and steps
        rng.nextPhotonId = (i * world + rank) * per_step
        photons.currentPhoton = 1
        integ.resetMoments()
        integ.computeRadiativeTransfer(dom, rng, photons, ppb, nb)
        if dist is not None:
            if a.pipeline:  # order the all-reduce after this step's moments and the next reset after the all-reduce
                integ.streamWaitDone(torch_stream)
            dist.all_reduce(moments, op=dist.ReduceOp.SUM)  # sumAcrossProcesses, monteCarloDriver.f95:1151-1166
            if a.pipeline:
                integ.waitStream(torch_stream)
            else:
                torch.cuda.synchronize()
        return 0.0 if a.pipeline else integ.lastTraceMs()

    def sync():
        integ.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if a.pipeline:
        integ.setAsync(True)  # (the first call still runs synchronously: it picks the event threshold by trial launches)
    for i in range(a.warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    for i in range(a.steps):
        kernel_ms += step(a.warmup + i)
    sync()
    elapsed = time.perf_counter() - t0
    if a.pipeline:
        kernel_ms = integ.lastTraceMs()  # summed over the timed steps (read at the synchronisation above)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        # the moment arrays now hold the last step: world * per_step photons, reduced over ranks
        stats = driver.statistics(driver.unpack_moments(moments.cpu().numpy(), nx, ny, nz))
        # untimed: event counters (instrumented kernel) on one step's worth of photons
        integ.bindMoments(0)
        if not a.no_cpu_baseline:
            # untimed parity run: 1e8 photons (BASELINE.json's accuracy target is quoted at 1e8)
            integ.resetMoments()
            rng.nextPhotonId = 10 ** 12
            photons.currentPhoton = 1
            integ.computeRadiativeTransfer(dom, rng, photons, ppb, a.parity_photons // ppb)
            stats = driver.statistics(driver.unpack_moments(integ.moments(), nx, ny, nz))
        integ.resetMoments()
        integ.enableCounters(True)
        rng.nextPhotonId = 0
        photons.currentPhoton = 1
        integ.computeRadiativeTransfer(dom, rng, photons, ppb, nb)
        cnt = integ.counters()
        integ.enableCounters(False)
        bpp = algorithmic_bytes_per_photon(cnt, per_step, nc)
        launch_ms = kernel_ms / a.steps
        achieved = bpp * per_step / (launch_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                rec = json.load(f).get(a.workload, {})
            if rec.get("hbm_bytes_per_launch") is not None:  # counters were collected on launches of photons_per_launch photons
                traffic = rec["hbm_bytes_per_launch"] * per_step / rec.get("photons_per_launch", per_step)
        out = {
            "metric": "photons/sec", "value": world * per_step * a.steps / elapsed, "unit": "photons/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32/f64",
            "data": "synthetic",
            "config": {"workload": "%s %dx%dx%d, %d photons/GPU/step as %d batches x %d, omega0=0.99 HG g=0.85 mu0=%g"
                       % (a.workload, nx, ny, nz, per_step, nb, ppb, w["mu0"]),
                       "photons_per_step_per_gpu": per_step, "parallelism": "photon batches sharded over %d GPU(s)" % world,
                       "pipelined_steps": bool(a.pipeline), "event_threshold": integ.eventThreshold()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_photon": bpp, "kernel": "trace_kernel", "kernel_ms_per_launch": launch_ms,
                         "events_per_photon": {k: v / per_step for k, v in cnt.items() if k in ("legs", "crossings", "collisions", "absorbEvents", "topExits", "surfaceHits", "rouletteKills", "rouletteSurvivals")},
                         "lanes_per_walk_iteration": cnt["walkLanes"] / max(1, cnt["walkIterations"]),
                         "lanes_per_event_phase": cnt["eventLanes"] / max(1, cnt["eventPhases"]),
                         "note": "working set is cache resident; the path is latency/VALU bound (DESIGN.md)"
                                 + ("; --pipeline: kernel durations include waiting for compute units held by the previous launch" if a.pipeline else "")},
        }
        if not a.no_cpu_baseline:
            cb, batches, cols, ccnt, ctot = cpu_baseline(a.workload, a.cpu_photons_per_core, a.cpu_cores)
            out["cpu_baseline"] = cb
            # flux error vs the CPU reference path, in units of the combined Monte Carlo sigma
            from oracle import oracle as O
            m_ref, e_ref = O.batch_statistics(batches)
            c_ref, ce_ref = O.batch_statistics(cols)
            g = np.array([stats["meanFluxUp"], stats["meanFluxDown"], stats["meanFluxAbsorbed"]])
            ge = np.array([stats["meanFluxUp_StdErr"], stats["meanFluxDown_StdErr"], stats["meanFluxAbsorbed_StdErr"]])
            z = (g - m_ref) / np.sqrt(ge ** 2 + e_ref ** 2 + 1e-30)
            gc = np.concatenate([stats[k].T.reshape(-1) for k in ("fluxUp", "fluxDown", "fluxAbsorbed")])
            gce = np.concatenate([stats[k + "_StdErr"].T.reshape(-1) for k in ("fluxUp", "fluxDown", "fluxAbsorbed")])
            zc = (gc - c_ref) / np.sqrt(gce ** 2 + ce_ref ** 2 + 1e-30)
            out["parity"] = {"gpu_photons": int(stats["totalPhotons"]), "cpu_photons": int(ctot),
                             "max_abs_z_domain_mean": float(np.max(np.abs(z))),
                             "max_abs_z_column": float(np.max(np.abs(zc))), "mean_z_column": float(np.mean(zc)),
                             "std_z_column": float(np.std(zc)), "frac_abs_z_column_gt_3": float(np.mean(np.abs(zc) > 3)),
                             "n_column_bins": int(zc.size),
                             "gpu_means": [float(x) for x in g], "cpu_means": [float(x) for x in m_ref],
                             "cpu_events_per_photon": {k: v / ctot for k, v in ccnt.items() if k != "draws"}}
        if world == 1 and a.workload == "i3rcStepCloud" and not a.no_secondary:
            out["secondary"] = secondary_workload(M, new_RandomNumberSequence)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
