#!/usr/bin/env python3
"""bench.py -- photons/sec of the photon-tracing hot path on N MI355X of one node.

    python bench.py                       # N = 1, finishes in about a minute
    python bench.py --gpus N              # starts its own N ranks (torch.distributed.run as a child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W      # or: launched by the driver

One "step" = one complete job of the workload on every rank: the I3RC step cloud
(BASELINE.json configs[1]: 32x1x32, tau 2|18, omega0 0.99, HG g 0.85, mu0 1) traced with
1e7 photons per GPU as 100 batches of 1e5 (computeRadiativeTransfer + reportResults +
batch moments on the device), followed by the all-reduce of the moment arrays over RCCL
(the reference's sumAcrossProcesses, Drivers/monteCarloDriver.f95:1151-1166).  Weak scaling:
every rank traces its own 1e7 photons per step (disjoint photon-id ranges of one Philox key).
Inputs (grids, tables) are resident in HBM before the timed region; photons are generated on
the GPU.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline     -- algorithmic bytes of the tracing kernel / its HIP-event duration vs HBM peak, the
                  HBM traffic the counters saw, and -- because instruction issue, not HBM, is what
                  binds this path -- a `valu` object (issue fraction, lane occupancy, wait split)
  cpu_baseline -- the CPU oracle (a C restatement of the reference loop: "port") timed on this
                  box's host cores on a bounded photon sample, with the measured speed ratio to the
                  reference Fortran (calibration_r) and the reference-equivalent rate
  parity       -- z-scores of the GPU's domain means, per-column fluxes and per-level heating
                  against that CPU sample (oracle in MT mode: the reference's generator and draw order)
  secondary    -- the 128x128x64 cloud field (the domain BASELINE.json's target is quoted on)

--dry-run rehearses the launcher and the collective on CPU (gloo): no tracing, no GPU.
"""
import argparse
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tests import cases  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.0  # wave64 VALU instructions per second: 1024 SIMD-32s, one per 2 cycles, 2.4 GHz
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9        # cycles of all 1024 SIMDs per second
WORKLOADS = {
    # cpu: photons per host core of the cpu_baseline sample (about 20 s of oracle time), parity: photons of the GPU parity run
    "i3rcStepCloud": dict(make=lambda: cases.step_cloud(ssa=0.99), mu0=1.0, phi0=0.0, ppb=100000, batches=100,
                          cpu=8000000, parity=400000000),
    "landsatLike128": dict(make=lambda: cases.landsat_like(), mu0=0.5, phi0=30.0, ppb=1000000, batches=100,
                           cpu=2000000, parity=100000000),
    # config 5: optically thick, omega0 = 0.9, roulette-heavy; not a headline line, kept for its parity record
    "radarLike128": dict(make=lambda: cases.radar_like(), mu0=0.5, phi0=30.0, ppb=1000000, batches=100,
                         cpu=4000000, parity=100000000),
    # config 4: broadband thermal emission, homogeneous isothermal 20x20x20, 16 wavelengths 8-12 um, every
    # wavelength's optics and emission CDF resident on the device, 1e8 photons per step split over the wavelengths on the device
    "homogLW20x16": dict(kind="lw", ppb=1000000, batches=100, cpu=8000000),
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
    means, cols, prof, counters, done = [], [], [], None, 0
    while done < n:
        nb = min(CPU_BATCH, n - done)
        res = O.compute_radiative_transfer(P, src, rng, nb)
        means.append((nb, np.array([res["meanFluxUp"], res["meanFluxDown"], res["meanFluxAbsorbed"]], np.float64)))
        cols.append((nb, np.concatenate([res["fluxUp"], res["fluxDown"], res["fluxAbsorbed"]]).astype(np.float64)))
        prof.append((nb, np.asarray(res["absorbedProfile"], np.float64)))
        counters = res["counters"] if counters is None else {k: counters[k] + v for k, v in res["counters"].items()}
        done += nb
    return time.time() - t, n, means, cols, counters, prof


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
    prof = [b for o in out for b in o[5]]
    counters = {k: sum(o[4][k] for o in out) for k in out[0][4]}
    cb = dict(value=total / busiest, unit="photons/s", cores=cores, kind="port",
              sample="%d photons/core x %d cores of the same workload in batches of %d, oracle in MT mode "
                     "(%.1f s wall)" % (photons_per_core, cores, CPU_BATCH, wall))
    cal = os.path.join(ROOT, "profiles", "cpu_calibration.json")
    if os.path.exists(cal):  # oracle vs reference Fortran, one core, same container (provenance in the file)
        with open(cal) as f:
            rec = json.load(f)
        cb["calibration_r"] = rec["r"]
        cb["reference_equivalent"] = cb["value"] / rec["r"]
        cb["calibration_source"] = rec["source"]
    return cb, batches, cols, counters, total, prof


def z_scores(g, ge, r, re):
    return (np.asarray(g) - np.asarray(r)) / np.sqrt(np.asarray(ge) ** 2 + np.asarray(re) ** 2 + 1e-30)


def parity_block(stats, batches, cols, prof, ccnt, ctot, flux=1.0):
    """z = (GPU - REF) / sqrt(sigma_GPU^2 + sigma_REF^2), sigma from the batch variance
    (the driver's estimator, monteCarloDriver.f95:1188-1219), over the domain means, every column flux and every
    level of the absorption (heating) profile; pass = max |z| < max(4, sqrt(2 ln N) + 1) over N bins (the largest of N
    unit normals grows like sqrt(2 ln N)) and |mean z| < 0.2."""
    from oracle import oracle as O
    m_ref, e_ref = O.batch_statistics(batches, solar_flux=flux)
    c_ref, ce_ref = O.batch_statistics(cols, solar_flux=flux)
    p_ref, pe_ref = O.batch_statistics(prof, solar_flux=flux)
    g = np.array([stats["meanFluxUp"], stats["meanFluxDown"], stats["meanFluxAbsorbed"]])
    ge = np.array([stats["meanFluxUp_StdErr"], stats["meanFluxDown_StdErr"], stats["meanFluxAbsorbed_StdErr"]])
    z = z_scores(g, ge, m_ref, e_ref)
    gc = np.concatenate([stats[k].T.reshape(-1) for k in ("fluxUp", "fluxDown", "fluxAbsorbed")])
    gce = np.concatenate([stats[k + "_StdErr"].T.reshape(-1) for k in ("fluxUp", "fluxDown", "fluxAbsorbed")])
    zc = z_scores(gc, gce, c_ref, ce_ref)
    live = (np.asarray(stats["absorbedProfile_StdErr"]) > 0) | (pe_ref > 0)  # (levels nothing is absorbed in carry no statistic)
    zp = z_scores(stats["absorbedProfile"], stats["absorbedProfile_StdErr"], p_ref, pe_ref)[live]
    lim_c = max(4.0, float(np.sqrt(2.0 * np.log(max(zc.size, 2))) + 1.0))
    ok = bool(np.max(np.abs(zc)) < lim_c and abs(np.mean(zc)) < 0.2 and (zp.size == 0 or np.max(np.abs(zp)) < 4.0)
              and np.max(np.abs(z)) < 4.0)
    return {"gpu_photons": int(stats["totalPhotons"]), "cpu_photons": int(ctot),
            "max_abs_z_domain_mean": float(np.max(np.abs(z))), "z_domain_means": [float(x) for x in z],
            "max_abs_z_column": float(np.max(np.abs(zc))), "mean_z_column": float(np.mean(zc)),
            "std_z_column": float(np.std(zc)), "frac_abs_z_column_gt_3": float(np.mean(np.abs(zc) > 3)),
            "n_column_bins": int(zc.size),
            "max_abs_z_heating_level": float(np.max(np.abs(zp))) if zp.size else 0.0,
            "mean_z_heating_level": float(np.mean(zp)) if zp.size else 0.0,
            "std_z_heating_level": float(np.std(zp)) if zp.size else 0.0, "n_heating_levels": int(zp.size),
            "thresholds": {"max_abs_z_bins": lim_c, "abs_mean_z": 0.2, "max_abs_z_levels": 4.0}, "within_thresholds": ok,
            "gpu_means": [float(x) for x in g], "cpu_means": [float(x) for x in m_ref],
            "cpu_events_per_photon": {k: v / ctot for k, v in ccnt.items() if k != "draws"}}


def pmc_record(workload):
    """Counter figures of the shipped kernel for this workload (profiles/pmc_shipped.json, written by
    scripts/pmc_summary.py from separate rocprofv3 --pmc passes; per launch of photons_per_launch photons)."""
    path = os.path.join(ROOT, "profiles", "pmc_shipped.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        rec = json.load(f).get(workload, {})
    if rec:
        # the counters were collected on another box at another time: they describe THIS library only if they were taken from
        # the same kernel sources (scripts/pmc_summary.py stores their sha256).  A mismatch does not drop the figures -- they are
        # still the best description on file -- but marks them and withholds the conclusion drawn from them (binding_resource).
        from mcbrat3d_amd.build import sources_sha256
        rec = dict(rec)
        rec["stale"] = rec.get("kernel_sources_sha256") != sources_sha256()
    return rec


def kernel_name(walk, thermal=False, lds_grid=None):
    """The tracing kernel a flux run of this walk mode launches (mcbrat_api.hip: launch_trace)."""
    if walk.get("blockWalk"):
        if walk.get("widePlan"):
            return "trace_block_kernel (mcbrat_blockwalk.hip: block walk, one workgroup of 1024 lanes per CU, tallies + tables%s in LDS%s)" % (
                " + block numbers, optics in global memory" if walk.get("opticsInGlobalMemory") else (" + block numbers and optics per block" if walk.get("opticsPerBlock") else " + grid"),
                (", thermal source" + (" with the emission CDF's level and row sums in LDS" if walk.get("emissionCdfTopInLds") else "")) if thermal else "")
        return "trace_block_kernel (mcbrat_blockwalk.hip: block walk, grid + tallies + tables in LDS%s)" % (", thermal source" if thermal else "")
    return "trace_kernel (mcbrat_kernels.hip: %s%s%s)" % (
        "layer-skipping walk" if walk.get("layerSkip") else "face-by-face walk",
        " + clear-air flight" if walk.get("clearAirFlight") else "", ", thermal source" if thermal else "")


def roofline_block(workload, cnt, per_step, nc, launch_ms, pipeline=False, block_walk=True, walk=None, thermal=False):
    bpp = algorithmic_bytes_per_photon(cnt, per_step, nc)
    achieved = bpp * per_step / (launch_ms * 1e-3) / 1e9
    # the walks that skip faces (block walk, layer skipping, clear-air flight) never stop at the cell faces whose 4 bytes
    # the algorithmic figure counts: the same rate on the bytes such a kernel can touch at all
    skips = bool(walk and (walk.get("blockWalk") or walk.get("layerSkip")))
    touched = bpp - (4.0 * cnt["crossings"] / per_step if skips else 0.0)
    rec = pmc_record(workload)
    if workload == "i3rcStepCloud" and not block_walk:
        rec = {}  # (the counters on file are the block-walk kernel's: they do not describe the face-by-face run)
    scale = per_step / rec.get("photons_per_launch", per_step)
    traffic = rec["hbm_bytes_per_launch"] * scale if rec.get("hbm_bytes_per_launch") is not None else None
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": traffic, "counters_stale": bool(rec.get("stale")) if rec else None, "achieved_is": "algorithmic bytes (SURVEY.md 8d) / kernel time: an equivalent rate, not HBM utilisation",
           "measured_hbm_GBps": (traffic / (launch_ms * 1e-3) / 1e9) if traffic is not None else None,
           "frac_on_touched_bytes": touched * per_step / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "touched_bytes_per_photon": touched,
           "algorithmic_bytes_per_photon": bpp, "kernel": kernel_name(walk or {"blockWalk": block_walk and workload == "i3rcStepCloud"}, thermal),
           "kernel_ms_per_launch": launch_ms,
           "events_per_photon": {k: v / per_step for k, v in cnt.items() if k in (
               "legs", "crossings", "collisions", "absorbEvents", "topExits", "surfaceHits", "rouletteKills", "rouletteSurvivals")},
           "lanes_per_walk_iteration": cnt["walkLanes"] / max(1, cnt["walkIterations"]),
           "lanes_per_event_phase": cnt["eventLanes"] / max(1, cnt["eventPhases"]),
           "binding_resource": ("valu_issue" if (workload == "i3rcStepCloud" or rec.get("wave_time_waiting", 1.0) < 0.4) else "l2_requests") if (rec and not rec.get("stale")) else None,
           "note": "working set is cache / LDS resident: HBM is not what binds (see valu); DESIGN.md section 5"
                   + ("; --pipeline: kernel durations include waiting for compute units held by the previous launch" if pipeline else "")}
    if rec.get("valu_insts_per_launch"):
        # VALU issue fraction from THIS run's kernel time and the instruction count the counters saw for the same launch shape
        insts = rec["valu_insts_per_launch"] * scale
        out["valu"] = {"issue_frac": insts / (launch_ms * 1e-3) / VALU_ISSUE_PEAK, "peak_wave_insts_per_s": VALU_ISSUE_PEAK,
                       "valu_wave_insts_per_photon": insts / per_step, "lane_occupancy": rec.get("lane_occupancy"),
                       "useful_lane_cycle_frac": (insts / (launch_ms * 1e-3) / VALU_ISSUE_PEAK) * (rec.get("lane_occupancy") or 0.0),
                       "wave_time_issuing": rec.get("wave_time_issuing"), "wave_time_waiting": rec.get("wave_time_waiting"),
                       "wave_time_issue_stall": rec.get("wave_time_issue_stall"),
                       "lds_bank_conflict_ratio": rec.get("lds_bank_conflict_ratio"),
                       "l2_requests_per_photon": (rec.get("l2_requests_per_launch") or 0.0) / rec.get("photons_per_launch", per_step),
                       "l2_hit_rate": rec.get("l2_hit_rate"), "source": rec.get("source"),
                       "stale": bool(rec.get("stale")), "kernel_sources_sha256": rec.get("kernel_sources_sha256")}
        if rec.get("valu_simd_cycles_per_launch"):
            # the same with every instruction class at the SIMD cycles it was measured to cost on this part (f64 4.3, Philox's 64-bit
            # multiplies 4.5, transcendentals 8.2 against 2.3 for plain f32 / int32: scripts/valu_rates.hip): the share of the
            # launch's SIMD cycles in which a vector instruction of this kernel was executing -- the roofline that binds it
            out["valu"]["simd_busy_frac"] = rec["valu_simd_cycles_per_launch"] * scale / (launch_ms * 1e-3 * SIMD_CYCLES_PER_S)
            out["valu"]["simd_cycles_per_valu_inst"] = rec["valu_simd_cycles_per_launch"] / rec["valu_insts_per_launch"]
    return out


XGMI_LINK_BYTES_PER_S = 153e9  # one xGMI link of an MI355X (7 per GPU), /opt/skills/guides/MI355X_MICROARCH.md


def one_rank_allreduce_ms(torch, tensor, device, reps=5):
    """Duration (CUDA events on torch's stream) of dist.all_reduce(tensor) in a ONE-rank RCCL group: what the collective
    costs before any byte crosses a link.  Only where no process group exists yet (the N = 1 bench); None if RCCL does not
    come up -- the projection then carries the link term alone."""
    try:
        import torch.distributed as dist
        if dist.is_initialized():
            return None
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(free_port())
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)  # (RCCL's banner must not reach stdout: the contract is one JSON line there)
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", device))
            probe = torch.zeros_like(tensor)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            ms = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dist.all_reduce(probe)
                e1.record()
                torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1))
            dist.destroy_process_group()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        return float(sorted(ms)[len(ms) // 2])
    except Exception as exc:  # noqa: BLE001
        print("bench.py: one-rank all-reduce probe failed (%s): projection without its fixed cost" % exc, file=sys.stderr)
        return None


def secondary_workload(M, new_rng, device, dist, rank, world, name="landsatLike128", steps=3):
    """Not the contract's metric: BASELINE.json's target is quoted on a 128x128x64 domain, so every run also reports
    that workload's whole-job rate: `steps` synchronous steps of 1e8 photons per GPU (each followed by the all-reduce
    of the moments when there are several ranks) after one warm-up step, which also lets the library choose its
    event threshold."""
    import torch
    w = WORKLOADS[name]
    dom = cases.product_domain(w["make"]())
    integ = M.new_Integrator(dom, device=device)
    integ.specifyParameters(minInverseTableSize=10001, useRayTracing=True, useRussianRoulette=True)
    photons = M.new_PhotonStream(w["mu0"], w["phi0"], numberOfPhotons=10 ** 15)
    rng = new_rng(10)
    per_step = w["ppb"] * w["batches"]
    dev = torch.device("cuda", device)
    moments = torch.zeros(8 + 2 * integ.momentsLength(), dtype=torch.float64, device=dev)
    integ.bindMoments(moments.data_ptr())

    def step(i):
        rng.nextPhotonId = (i * world + rank) * per_step
        photons.currentPhoton = 1
        integ.resetMoments()
        integ.computeRadiativeTransfer(dom, rng, photons, w["ppb"], w["batches"])
        if dist is not None:
            dist.all_reduce(moments, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
        return integ.lastTraceMs()

    step(0)
    integ.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    t0, kms = time.perf_counter(), 0.0
    for i in range(steps):
        kms += step(1 + i)
    integ.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # Strong scaling (north_star: ">= 6x scaling 1 -> 8 GPUs at 1e8 photons"): ONE job of per_step photons whose batches
    # are split over the ranks as the reference's driver splits its work units (monteCarloDriver.f95:665-880), moments
    # all-reduced at the end.  With one rank the same job cut into the per-GPU shares of 2, 4 and 8 ranks is timed launch
    # by launch: a launch ends with its longest photon history, so a smaller share runs at a lower rate, and the ratio of
    # the two launch times is the speed-up that share would give (the all-reduce of 17 MB over xGMI adds about a millisecond).
    from mcbrat3d_amd import driver as _driver
    strong = None

    def strong_step(i, share_of):
        # (the job cut into a batch count that is a multiple of the rank count: every rank the same number of whole batches)
        ppb_n, nb_n = _driver.balanced_job(per_step, w["batches"], share_of)
        lo, mine = _driver.split_batches(nb_n, rank if share_of == world else 0, share_of)
        rng.nextPhotonId = (10 ** 6 + i) * per_step + lo * ppb_n
        photons.currentPhoton = 1
        integ.resetMoments()
        integ.computeRadiativeTransfer(dom, rng, photons, ppb_n, mine)
        if dist is not None:
            dist.all_reduce(moments, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
        return integ.lastTraceMs()

    def share_photons(n):
        ppb_n, nb_n = _driver.balanced_job(per_step, w["batches"], n)
        return _driver.split_batches(nb_n, 0, n)[1] * ppb_n

    if world > 1:
        strong_step(0, world)
        dist.barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(steps):
            strong_step(1 + i, world)
        dist.barrier(); torch.cuda.synchronize()
        dts = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(dts, op=dist.ReduceOp.MAX)
        strong = {"job_photons": per_step, "n_gpus": world, "value": per_step * steps / float(dts.item()), "unit": "photons/s",
                  "ms_per_job": 1e3 * float(dts.item()) / steps}
    else:
        shares = {}
        for n in (1, 2, 4, 8):
            strong_step(0, n)  # warm
            t1 = time.perf_counter()
            km = sum(strong_step(1 + i, n) for i in range(2))
            shares[n] = {"photons_per_launch": share_photons(n), "ms_per_launch": 1e3 * (time.perf_counter() - t1) / 2, "kernel_ms": km / 2}
        t1ms = shares[1]["ms_per_launch"]
        # the collective the projection leaves out, priced: the all-reduce of THIS moment array through a one-rank RCCL group,
        # measured here (launch + one pass over the array), plus what the ring moves over xGMI with N ranks --
        # 2 (N-1)/N x bytes over one link of ~153 GB/s (MI355X_MICROARCH: xGMI is point to point, a ring is bound per link)
        ar1 = one_rank_allreduce_ms(torch, moments, device)
        nbytes = moments.numel() * moments.element_size()
        ar = {n: (ar1 if ar1 is not None else 0.0) + 1e3 * 2.0 * (n - 1) / n * nbytes / XGMI_LINK_BYTES_PER_S for n in (2, 4, 8)}
        strong = {"job_photons": per_step, "n_gpus": 1, "per_gpu_share_timed_on_one_gpu": shares,
                  "projected_speedup": {str(n): t1ms / shares[n]["ms_per_launch"] for n in (2, 4, 8)},
                  "projected_speedup_with_allreduce": {str(n): t1ms / (shares[n]["ms_per_launch"] + ar[n]) for n in (2, 4, 8)},
                  "projected_photons_per_s": {str(n): per_step / (shares[n]["ms_per_launch"] * 1e-3) for n in (1, 2, 4, 8)},
                  "allreduce": {"bytes": nbytes, "one_rank_rccl_ms_measured": ar1, "xgmi_link_GBps_assumed": XGMI_LINK_BYTES_PER_S / 1e9,
                                "modelled_ms": {str(n): ar[n] for n in (2, 4, 8)}},
                  "note": "projection from one GPU: time of the share a rank would trace (1/N of the job -- cut into a batch count "
                          "that is a multiple of N, driver.balanced_job -- in one synchronous launch, finish kernels included); "
                          "projected_speedup leaves the all-reduce of the moment array out, projected_speedup_with_allreduce adds "
                          "the one-rank RCCL all-reduce measured here plus the ring's bytes over one xGMI link"}
    res = None
    if rank == 0:
        integ.bindMoments(0)
        integ.enableCounters(True)
        integ.computeRadiativeTransfer(dom, rng, photons, w["ppb"], 10)
        cnt = integ.counters()
        integ.enableCounters(False)
        rl = roofline_block(name, cnt, w["ppb"] * 10, len(dom.components), kms / steps * 0.1, walk=integ.walkMode())
        res = {"workload": "%s %dx%dx%d, %d photons/GPU/step" % (name, dom.numX, dom.numY, dom.numZ, per_step),
               "value": world * per_step * steps / dt, "unit": "photons/s", "n_gpus": world, "steps": steps,
               "ms_per_step": 1e3 * dt / steps, "kernel_ms_per_launch": kms / steps,
               "algorithmic_bytes_per_photon": rl["algorithmic_bytes_per_photon"], "roofline_frac": rl["frac"],
               "roofline": rl, "event_threshold": integ.eventThreshold(), "strong_scaling": strong,
               "bad_photons": integ.badPhotons()}
    integ.finalize()
    return res


def lw_bench(a, M, torch, dist, dev, rank, local_rank, world, rehearse):
    """--workload homogLW20x16 (BASELINE.json configs[3]): the reference's thermal broadband loop
    (monteCarloDriver.f95:304-449 set-up pass, :889-1085 worker loop; emission_weightingNEW
    emissionAndBroadBandWeights.f95:424-550, getFrequencyDistr :552-572, newPhotonStream_BBEmission
    monteCarloIllumination.f95:431-522) with every wavelength's optics, inverse tables and emission CDF resident on the
    device (mcbrat3d_amd.broadband.SpectralRun).  One step = 1e8 photons per GPU: the photon split over the 16
    wavelengths on the device, then every wavelength's photons in batches of 1e6; nothing is uploaded inside the
    timed loop.  Weak scaling over ranks; one all-reduce of the shared moment array per step."""
    from mcbrat3d_amd import broadband, driver
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    from tests import stats as tstats
    w = WORKLOADS[a.workload]
    cs = tstats.lw_cases()
    doms = [cases.product_domain(c) for c in cs]
    nx, ny, nz = doms[0].numX, doms[0].numY, doms[0].numZ
    t_setup = time.perf_counter()
    run = broadband.SpectralRun(M, doms, device=local_rank, minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=True)
    for it in run.integrators:
        it.setTuning(eventThreshold=a.event_threshold)
    flux = run.prepare_thermal(tstats.LW_SURFACE_TEMP)  # set-up pass + uploads, once, outside the timed region
    t_setup = time.perf_counter() - t_setup
    moments = torch.zeros(8 + 2 * run.first.momentsLength(), dtype=torch.float64, device=dev)
    run.bindMoments(moments.data_ptr())
    ppb, nb = w["ppb"], w["batches"]
    per_step = ppb * nb
    rng = new_RandomNumberSequence(10)
    launches = [0]

    def step(i):
        rng.nextPhotonId = (i * world + rank) * per_step
        run.resetMoments()
        counts = run.run(ppb, nb, rng, seed=1000 + i * world + rank)
        for it in run.integrators:  # (a step is synchronous: every wavelength's photons traced and folded into the moments)
            it.synchronize()
        kms = 0.0  # (the wavelengths' tracing kernels overlap: their durations are measured apart, below)
        launches[0] = int(sum((c // ppb) > 0 for c in counts) + sum((c % ppb) > 0 for c in counts))
        if dist is not None:
            dist.all_reduce(moments, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
        return kms

    def sync():
        for it in run.integrators:
            it.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    sync()
    from mcbrat3d_amd import _capi
    hip_runtime = _capi.assert_single_hip_runtime()  # (torch's tensor is the moment array the library writes: one runtime serves both)
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    out = None
    if rank == 0:
        stats = driver.statistics(driver.unpack_moments(moments.cpu().numpy(), nx, ny, nz), solarFlux=flux)
        reduced_photons = int(stats["totalPhotons"])
        if world > 1:  # parity is quoted on one rank's 1e8 photons
            run.bindMoments(0)
        # untimed: one step with every wavelength's photons in ONE call each: tracing-kernel time per step (HIP events around
        # each tracing kernel, summed over the 16 wavelengths) and, from the instrumented instantiation, the event counters
        run.set_overlap(False)  # (kernel durations that describe the kernel: one launch at a time, the host waits for each)

        def one_call_per_wavelength(counters):
            for it in run.integrators:
                it.enableCounters(counters)
            run.resetMoments()
            rng.nextPhotonId = 10 ** 12
            run.run(per_step, 1, rng, seed=77)
            return sum(it.lastTraceMs() for it in run.integrators)
        one_call_per_wavelength(False)
        launch_ms = one_call_per_wavelength(False)
        one_call_per_wavelength(True)
        cnt = None
        for it in run.integrators:
            c = it.counters()
            cnt = c if cnt is None else {k: cnt[k] + v for k, v in c.items()}
        for it in run.integrators:
            it.enableCounters(False)
        walk = run.first.walkMode()
        rl = roofline_block(a.workload, cnt, per_step, 1, launch_ms, walk=walk, thermal=True)
        rl["launches"] = "one per wavelength (16): kernel_ms_per_launch is their sum for one step's 1e8 photons"
        out = {"metric": "photons/sec", "value": world * per_step * a.steps / elapsed, "unit": "photons/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32/f64", "data": "synthetic",
               "config": {"workload": "%s: broadband thermal emission, %dx%dx%d homogeneous isothermal (280 K, surface 300 K, albedo 0.1, "
                                      "ext 5 /km, omega0 0.5, HG g 0.85), 16 wavelengths 8-12 um resident on the device, %d photons/GPU/step "
                                      "split over the wavelengths on the device, batches of %d" % (a.workload, nx, ny, nz, per_step, ppb),
                          "photons_per_step_per_gpu": per_step, "parallelism": "photon batches sharded over %d GPU(s)" % world,
                          "world_size": world, "photons_in_reduced_moments_last_step": reduced_photons,
                          "kernel_launches_per_step": launches[0], "setup_and_upload_s_once": t_setup,
                          "wavelengths_overlap": "tracing kernels of consecutive wavelengths overlap on the GPU (each context asynchronous, "
                                                 "finish chains ordered on the device by mcbrat_chain_after); roofline.kernel_ms_per_launch is "
                                                 "measured apart with one launch at a time",
                          "emitted_flux_W_m2": flux, "event_threshold": run.first.eventThreshold(), "walk": walk,
                          "bad_photons": int(sum(it.badPhotons() for it in run.integrators)),
                          "hip_runtime": hip_runtime,
                          **({"rehearsal": "gloo, all ranks on cuda:0 -- not a measurement"} if rehearse else {})},
               "roofline": rl}
        if not a.no_cpu_baseline:
            cores = max(1, min(a.cpu_cores, len(os.sched_getaffinity(0))))
            n_core = a.cpu_photons_per_core or w["cpu"]
            t = time.time()
            rows, cflux, ccnt, busy = tstats.oracle_lw_run(n_core, CPU_BATCH, cores)
            ctot = n_core * cores
            cb = dict(value=ctot / busy, unit="photons/s", cores=cores, kind="port",
                      sample="%d photons/core x %d cores of the same workload (every core the whole spectrum, its own MT stream: "
                             "wavelength per photon by getFrequencyDistr, batches of %d), oracle in MT mode (%.1f s wall)"
                             % (n_core, cores, CPU_BATCH, time.time() - t))
            cal = os.path.join(ROOT, "profiles", "cpu_calibration.json")
            if os.path.exists(cal):
                with open(cal) as f:
                    rec = json.load(f)
                cb["calibration_r"] = rec["r"]
                cb["reference_equivalent"] = cb["value"] / rec["r"]
                cb["calibration_source"] = rec["source"] + " (measured on the step cloud; applied to this workload as a proxy)"
            out["cpu_baseline"] = cb
            if world > 1:
                run.set_overlap(True)
                run.resetMoments(); rng.nextPhotonId = 5 * 10 ** 12
                run.run(ppb, nb, rng, seed=5)
                stats = driver.statistics(driver.unpack_moments(run.moments(), nx, ny, nz), solarFlux=flux)
            assert abs(cflux - flux) <= 1e-9 * abs(flux), (cflux, flux)  # both sides built the same spectrum
            out["parity"] = parity_block(stats, rows["means"], rows["columns"], rows["profile"], ccnt, ctot, flux=flux)
    run.finalize()
    if rank == 0:
        print(json.dumps(out), flush=True)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside a launcher: start N ranks as a CHILD process (torch.distributed.run, one
    process per GPU) and pass its output through.  Nothing in this process has touched the GPU -- torch is not even
    imported yet -- and nothing is exec'ed: the child's exit code is returned."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    env["BENCH_LAUNCHED"] = "1"
    return subprocess.call(cmd, env=env)


def dry_run(a, rank, world):
    """Launcher / collective rehearsal on CPU: gloo, no tracing, no GPU.  Every rank contributes a moments-shaped
    array with its rank in it; rank 0 prints ONE JSON line with the ranks it saw."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(free_port()))
    sys.stdout.flush()
    saved_fd = os.dup(1)  # (gloo announces its connections on stdout; the contract is ONE JSON line there)
    os.dup2(2, 1)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    finally:
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    w = WORKLOADS[a.workload]
    per_step = w["ppb"] * w["batches"]
    seen = torch.zeros(world, dtype=torch.float64)
    seen[rank] = 1.0 + rank
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.warmup + a.steps):
        buf = torch.zeros(8 + 2 * 16, dtype=torch.float64)
        buf[0], buf[1] = per_step, w["batches"]
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)  # sumAcrossProcesses
    dist.all_reduce(seen, op=dist.ReduceOp.SUM)
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "photons/sec", "value": 0.0, "unit": "photons/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": 1e3 * float(tt.item()) / max(1, a.steps), "higher_is_better": True,
                          "scaling": a.scaling, "vs_baseline": None, "dtype": "f32/f64", "data": "synthetic", "dry_run": True,
                          "config": {"workload": a.workload, "world_size": dist.get_world_size(), "backend": "gloo",
                                     "ranks_seen": [int(x) - 1 for x in seen.tolist()],
                                     "photons_all_ranks_per_step": int(buf[0].item())}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="i3rcStepCloud", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-photons-per-core", type=int, default=0, help="0 = the workload's default")
    ap.add_argument("--cpu-cores", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the untimed 128x128x64 extra")
    ap.add_argument("--parity-photons", type=int, default=0, help="0 = the workload's default")
    ap.add_argument("--event-threshold", type=int, default=0,
                    help="0 = let the library time trial launches (default); >0 fixes it (profiling runs)")
    ap.add_argument("--block-walk", type=int, default=-1, help="-1 library default; 0 face-by-face walk; 1 block walk")
    ap.add_argument("--pipeline", action="store_true",
                    help="let consecutive steps overlap on the GPU (asynchronous mode of the library: hides each "
                         "launch's drain; per-kernel durations then include time spent waiting for compute units, so "
                         "the default run, whose kernel durations rocprofv3 must reproduce, does not use it)")
    ap.add_argument("--pipelined-extra", action="store_true",
                    help="after the timed steps, run them once more with overlapping calls and report that rate as an "
                         "untimed extra in config (not in the default run: its kernel launches would enter a profiler's averages)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default, the driver's contract): every rank traces the workload's photons per step; strong: the "
                         "step's photons are one fixed job whose batches are split over the ranks (monteCarloDriver.f95:665-880)")
    ap.add_argument("--dry-run", action="store_true", help="rehearse launcher + collective on CPU (gloo), no tracing")
    a = ap.parse_args()

    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if a.gpus > 1 and not under_launcher:
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if under_launcher and a.gpus != world and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d ranks; reporting n_gpus = %d" % (a.gpus, world, world), file=sys.stderr)
    if a.dry_run:
        return dry_run(a, rank, world)
    if world > 1:
        a.no_cpu_baseline = True  # the CPU baseline (and the parity check against it) belongs to the N = 1 line
    import torch
    dist = None
    # BENCH_REHEARSE=1: every rank on cuda:0 and gloo instead of RCCL (which refuses two ranks on one device) -- the
    # whole N > 1 code path of this file on a one-GPU box; the numbers it prints are not a measurement
    rehearse = bool(os.environ.get("BENCH_REHEARSE"))
    if rehearse:
        local_rank = 0
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):  # BENCH_FORCE_DIST: exercise the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # BENCH_FORCE_DIST outside a launcher: a one-rank job
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        # RCCL prints a version banner on stdout when its communicator comes up; the contract is ONE JSON line there,
        # so file descriptor 1 points at stderr until the first collective is through
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearse:
                dist.init_process_group("gloo")
            else:
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
    if w.get("kind") == "lw":
        lw_bench(a, M, torch, dist, dev, rank, local_rank, world, rehearse)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    a.cpu_photons_per_core = a.cpu_photons_per_core or w["cpu"]
    a.parity_photons = a.parity_photons or w["parity"]
    case = w["make"]()
    dom = cases.product_domain(case)
    nx, ny, nz = dom.numX, dom.numY, dom.numZ
    nc = len(dom.components)
    integ = M.new_Integrator(dom, device=local_rank)
    integ.specifyParameters(minInverseTableSize=10001, useRayTracing=True, useRussianRoulette=True)
    integ.setTuning(eventThreshold=a.event_threshold, blockWalk=a.block_walk)
    photons = M.new_PhotonStream(w["mu0"], w["phi0"], numberOfPhotons=10 ** 15)
    ppb, nb = w["ppb"], w["batches"]
    per_step = ppb * nb
    moments = torch.zeros(8 + 2 * integ.momentsLength(), dtype=torch.float64, device=dev)
    integ.bindMoments(moments.data_ptr())
    rng = new_RandomNumberSequence(10)
    torch_stream = torch.cuda.current_stream(dev).cuda_stream

    # strong scaling: the job cut into a batch count that is a multiple of the rank count (every rank the same number of
    # whole batches; 100 x 1e6 on 8 ranks -> 104 x 961 538), weak scaling: the workload's own batches on every rank
    ppb_job, nb_job = driver.balanced_job(per_step, nb, world) if a.scaling == "strong" else (ppb, nb)
    job_photons = ppb_job * nb_job

    reduce_events = []  # (start, end) CUDA events around every all-reduce of the timed steps

    def step(i, timed=False):
        # weak scaling: rank r traces photon ids [ (i*world + r) * per_step, +per_step ): disjoint over ranks and steps
        # strong scaling: the step's per_step photons are cut into batches and rank r takes its contiguous share of them
        if a.scaling == "strong":
            lo, mine = driver.split_batches(nb_job, rank, world)
            rng.nextPhotonId = i * per_step + lo * ppb_job
        else:
            lo, mine = 0, nb
            rng.nextPhotonId = (i * world + rank) * per_step
        photons.currentPhoton = 1
        integ.resetMoments()
        if mine > 0:
            integ.computeRadiativeTransfer(dom, rng, photons, ppb_job, mine)
        if dist is not None:
            # the all-reduce is ordered after this step's moments and the next reset after the all-reduce ON THE DEVICE
            # (events): the host does not wait for the collective, the timed region ends with a synchronisation
            integ.streamWaitDone(torch_stream)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if timed else None
            if ev:
                ev[0].record()
            dist.all_reduce(moments, op=dist.ReduceOp.SUM)  # sumAcrossProcesses, monteCarloDriver.f95:1151-1166
            if ev:
                ev[1].record()
                reduce_events.append(ev)
            integ.waitStream(torch_stream)
        return 0.0 if (a.pipeline or mine == 0) else integ.lastTraceMs()

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
    # torch's stream and device pointers are handed to the library above: both must be served by ONE HIP runtime
    # (torch was imported first, so the library bound to torch's copy; mcbrat3d_amd/_capi.py: hip_runtimes)
    from mcbrat3d_amd import _capi
    hip_runtime = _capi.assert_single_hip_runtime()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    for i in range(a.steps):
        kernel_ms += step(a.warmup + i, timed=True)
    sync()
    elapsed = time.perf_counter() - t0
    if a.pipeline:
        kernel_ms = integ.lastTraceMs()  # summed over the timed steps (read at the synchronisation above)
    per_rank_kernel_ms, allreduce_ms = [kernel_ms / a.steps], None
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # diagnostics for the scaling curve: every rank's tracing-kernel time per step and the all-reduce's own duration
        # (CUDA events on the stream the collective runs on), so that a shortfall can be read off the line itself
        km = torch.tensor([kernel_ms / a.steps], dtype=torch.float64, device=dev)
        gathered = [torch.zeros_like(km) for _ in range(world)]
        dist.all_gather(gathered, km)
        per_rank_kernel_ms = [float(g.item()) for g in gathered]
        if reduce_events and not rehearse:
            allreduce_ms = sum(e0.elapsed_time(e1) for e0, e1 in reduce_events) / len(reduce_events)

    # Untimed extra (--pipelined-extra, N = 1): the same K steps with consecutive calls allowed to overlap on
    # the GPU (mcbrat_set_async).  A 1e7-photon launch ends with a tail in which most lanes wait for the last histories
    # (26 photons per lane); overlapping calls fills it.  Reported beside `value`, never as `value`: kernel durations
    # under overlap include time spent waiting for compute units, so they stop describing the kernel.
    pipelined_rate = None
    if world == 1 and not a.pipeline and a.pipelined_extra:
        integ.setAsync(True)
        t1 = time.perf_counter()
        for i in range(a.steps):
            rng.nextPhotonId = (a.warmup + a.steps + i) * per_step
            photons.currentPhoton = 1
            integ.resetMoments()
            integ.computeRadiativeTransfer(dom, rng, photons, ppb, nb)
        integ.synchronize()
        pipelined_rate = per_step * a.steps / (time.perf_counter() - t1)
        integ.setAsync(False)
        step(a.warmup + a.steps - 1)  # (the moment arrays hold the last timed step again)
        sync()

    out = None
    thr_timed = integ.eventThreshold()  # (a later, larger call -- the parity run -- may trigger the trial launches and choose another)
    if rank == 0:
        # the moment arrays now hold the last step: world * per_step photons, reduced over ranks
        stats = driver.statistics(driver.unpack_moments(moments.cpu().numpy(), nx, ny, nz))
        reduced_photons = int(stats["totalPhotons"])
        integ.bindMoments(0)
        if not a.no_cpu_baseline:
            # untimed parity run (BASELINE.json's accuracy target is quoted at 1e8 photons)
            integ.resetMoments()
            rng.nextPhotonId = 10 ** 12
            photons.currentPhoton = 1
            integ.computeRadiativeTransfer(dom, rng, photons, ppb, a.parity_photons // ppb)
            stats = driver.statistics(driver.unpack_moments(integ.moments(), nx, ny, nz))
        # untimed: event counters (instrumented kernel) on one step's worth of photons, at the timed steps' event threshold
        integ.setTuning(eventThreshold=thr_timed)
        integ.resetMoments()
        integ.enableCounters(True)
        rng.nextPhotonId = 0
        photons.currentPhoton = 1
        integ.computeRadiativeTransfer(dom, rng, photons, ppb, nb)
        cnt = integ.counters()
        integ.enableCounters(False)
        launch_ms = kernel_ms / a.steps
        out = {
            "metric": "photons/sec", "value": (job_photons if a.scaling == "strong" else world * per_step) * a.steps / elapsed, "unit": "photons/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f32/f64",
            "data": "synthetic",
            "config": {"workload": "%s %dx%dx%d, %d photons/%s/step as %d batches x %d, omega0=0.99 HG g=0.85 mu0=%g"
                       % (a.workload, nx, ny, nz, job_photons if a.scaling == "strong" else per_step, "job" if a.scaling == "strong" else "GPU",
                          nb_job, ppb_job, w["mu0"]),
                       "photons_per_step_per_gpu": job_photons // world if a.scaling == "strong" else per_step,
                       "parallelism": "photon batches sharded over %d GPU(s)" % world,
                       "kernel_ms_per_step_per_rank": per_rank_kernel_ms, "allreduce_ms_per_step": allreduce_ms,
                       "bad_photons": integ.badPhotons(),
                       "hip_runtime": hip_runtime,
                       "world_size": dist.get_world_size() if dist is not None else 1,
                       **({"rehearsal": "gloo, all ranks on cuda:0 -- not a measurement"} if rehearse else {}),
                       "photons_in_reduced_moments_last_step": reduced_photons,
                       "pipelined_steps": bool(a.pipeline), "pipelined_photons_per_s_untimed_extra": pipelined_rate,
                       "event_threshold": thr_timed,
                       "walk": integ.walkMode()},
            "roofline": roofline_block(a.workload, cnt, per_step, nc, launch_ms, a.pipeline, integ.walkMode()["blockWalk"], walk=integ.walkMode()),
        }
        if not a.no_cpu_baseline:
            cb, batches, cols, ccnt, ctot, prof = cpu_baseline(a.workload, a.cpu_photons_per_core, a.cpu_cores)
            out["cpu_baseline"] = cb
            out["parity"] = parity_block(stats, batches, cols, prof, ccnt, ctot)
    integ.finalize()
    if a.workload == "i3rcStepCloud" and not a.no_secondary:  # every rank takes part (all-reduce inside)
        sec = secondary_workload(M, new_RandomNumberSequence, local_rank, dist if world > 1 else None, rank, world)
        if rank == 0:
            out["secondary"] = sec
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
