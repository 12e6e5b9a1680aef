"""The N > 1 path on CPU: two gloo ranks shard the batches, keep per-rank batch moments,
all-reduce them, and must reproduce the single-rank statistics.  The per-rank tracing is
done here by the CPU oracle in Philox mode (a stand-in for the GPU of each rank; photon
ids, not generator state, carry the random numbers, so the split is exact)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mcbrat3d_amd import driver
from tests import cases

PPB, NB, SEED = 4000, 6, 99


def _moments_from_batches(P, src_args, first_batch, n_batches, nx, ny, nz, ppb=None):
    """What one rank's device holds after its batches: header + S1 + S2 (driver layout)."""
    from oracle import oracle as O
    PPB = ppb or globals()["PPB"]
    ncol, nvox = nx * ny, nx * ny * nz
    M = 3 + 3 * ncol + nz + nvox
    buf = np.zeros(8 + 2 * M)
    for b in range(first_batch, first_batch + n_batches):
        res = O.compute_radiative_transfer(P, O.solar_source(*src_args), O.philox_rng(SEED, b * PPB), PPB)
        x = np.concatenate([[res["meanFluxUp"], res["meanFluxDown"], res["meanFluxAbsorbed"]], res["fluxUp"],
                            res["fluxDown"], res["fluxAbsorbed"], res["absorbedProfile"],
                            res["volumeAbsorption"]]).astype(np.float64)
        buf[0] += PPB
        buf[1] += 1
        buf[8:8 + M] += PPB * x
        buf[8 + M:] += PPB * x * x
    return buf


class OracleIntegrator:
    """Stands where each rank's GPU integrator does in driver.run (tests only): same calls, traced by the oracle in
    Philox mode, batch moments accumulated into the bound buffer exactly as the finish kernels do."""

    def __init__(self, P, src_args, dims, buf):
        self.P, self.src_args, self._dims, self.buf, self.device = P, src_args, dims, buf, 0
        self.synchronised = 0

    def resetMoments(self):
        self.buf[:] = 0.0

    def synchronize(self):
        self.synchronised += 1

    def computeRadiativeTransfer(self, dom, rng, photons, ppb, nb):
        first = rng.nextPhotonId // ppb
        self.buf += _moments_from_batches(self.P, self.src_args, first, nb, *self._dims)
        rng.nextPhotonId += ppb * nb
        return ppb * nb

    def moments(self):
        return self.buf.copy()


def _worker_driver_run(rank, world, port, out):
    """driver.run itself over gloo: split, trace, all-reduce, statistics."""
    import mcbrat3d_amd as M
    from mcbrat3d_amd.integrator import RandomNumberSequence
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = cases.oracle_problem(cases.step_cloud(0.99))
    M_len = 3 + 3 * 32 + 32 + 32 * 32
    t = torch.zeros(8 + 2 * M_len, dtype=torch.float64)
    integ = OracleIntegrator(P, (1.0, 0.0), (32, 1, 32), t.numpy())
    rng = RandomNumberSequence(SEED, 0)
    photons = M.new_PhotonStream(1.0, 0.0, numberOfPhotons=PPB * NB)
    stats = driver.run(integ, None, photons, PPB, NB, rng, dist=dist, moments_tensor=t)
    assert integ.synchronised == 1  # (CPU tensors: the integrator is synchronised before the all-reduce)
    assert rng.nextPhotonId == PPB * NB
    if rank == 0:
        np.save(out, np.concatenate([[stats["totalPhotons"], stats["batches"], stats["meanFluxUp"], stats["meanFluxDown"],
                                      stats["meanFluxAbsorbed"], stats["meanFluxUp_StdErr"]], stats["absorbedProfile"]]))
    dist.barrier()
    dist.destroy_process_group()


def _worker_balanced(rank, world, port, out):
    """driver.balanced_job + split_batches over gloo: every rank the same number of whole batches."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = cases.oracle_problem(cases.step_cloud(0.99))
    ppb, nb = driver.balanced_job(PPB * 5, 5, world)  # 5 batches on 2 ranks -> 6 batches of 3333
    assert nb % world == 0
    lo, mine = driver.split_batches(nb, rank, world)
    assert mine == nb // world
    buf = torch.from_numpy(_moments_from_batches(P, (1.0, 0.0), lo, mine, 32, 1, 32, ppb=ppb))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(out, buf.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _worker_dynamic(rank, world, port, out):
    """driver.run(schedule="dynamic"): units of one batch dealt out from the shared counter; rank 1 is slow."""
    import time
    import mcbrat3d_amd as M
    from mcbrat3d_amd.integrator import RandomNumberSequence
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = cases.oracle_problem(cases.step_cloud(0.99))
    M_len = 3 + 3 * 32 + 32 + 32 * 32
    t = torch.zeros(8 + 2 * M_len, dtype=torch.float64)
    integ = OracleIntegrator(P, (1.0, 0.0), (32, 1, 32), t.numpy())
    calls = []
    trace = integ.computeRadiativeTransfer

    def slow(dom, rng, photons, ppb, nb):
        calls.append((rng.nextPhotonId // ppb, nb))
        if rank == 1:
            time.sleep(0.3)  # (a busy GPU: the other rank takes more units)
        return trace(dom, rng, photons, ppb, nb)
    integ.computeRadiativeTransfer = slow
    rng = RandomNumberSequence(SEED, 0)
    photons = M.new_PhotonStream(1.0, 0.0, numberOfPhotons=PPB * NB)
    stats = driver.run(integ, None, photons, PPB, NB, rng, dist=dist, moments_tensor=t, schedule="dynamic", unitBatches=1)
    assert rng.nextPhotonId == PPB * NB
    mine = torch.zeros(NB, dtype=torch.float64)
    for b, n in calls:
        assert n == 1
        mine[b] += 1
    dist.all_reduce(mine)
    assert bool((mine == 1).all()), mine  # every batch traced exactly once, by one rank
    counts = torch.zeros(world, dtype=torch.float64)
    counts[rank] = len(calls)
    dist.all_reduce(counts)
    if rank == 0:
        np.save(out, np.concatenate([[stats["totalPhotons"], stats["batches"], stats["meanFluxUp"], stats["meanFluxDown"],
                                      stats["meanFluxAbsorbed"], stats["meanFluxUp_StdErr"]], stats["absorbedProfile"], counts.numpy()]))
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = cases.step_cloud(0.99)
    P = cases.oracle_problem(case)
    lo, nb = driver.split_batches(NB, rank, world)
    buf = torch.from_numpy(_moments_from_batches(P, (1.0, 0.0), lo, nb, 32, 1, 32))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)  # sumAcrossProcesses
    if rank == 0:
        np.save(out, buf.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_split_batches_covers_everything():
    for nb in (1, 5, 8, 100):
        for world in (1, 2, 3, 8):
            parts = [driver.split_batches(nb, r, world) for r in range(world)]
            assert sum(n for _, n in parts) == nb
            pos = 0
            for lo, n in parts:
                assert lo == pos
                pos += n


def test_driver_run_over_two_gloo_ranks(tmp_path):
    out = str(tmp_path / "stats.npy")
    mp.spawn(_worker_driver_run, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    P = cases.oracle_problem(cases.step_cloud(0.99))
    single = driver.statistics(driver.unpack_moments(_moments_from_batches(P, (1.0, 0.0), 0, NB, 32, 1, 32), 32, 1, 32))
    want = np.concatenate([[single["totalPhotons"], single["batches"], single["meanFluxUp"], single["meanFluxDown"],
                            single["meanFluxAbsorbed"], single["meanFluxUp_StdErr"]], single["absorbedProfile"]])
    assert np.allclose(got, want, rtol=1e-12, atol=1e-15)


def test_balanced_job_gives_every_rank_the_same_share():
    for total, nb, world in ((10 ** 8, 100, 8), (10 ** 8, 100, 1), (10 ** 7, 100, 3), (1000, 7, 4), (5, 3, 8)):
        ppb, n = driver.balanced_job(total, nb, world)
        assert n % world == 0 and n >= nb and n - nb < world and ppb >= 1
        shares = [driver.split_batches(n, r, world)[1] * ppb for r in range(world)]
        assert len(set(shares)) == 1
        assert abs(ppb * n - total) <= n / 2 + 1 or ppb == 1  # the job keeps its size to within half a photon per batch
    assert driver.balanced_job(10 ** 8, 100, 8) == (961538, 104)  # 12.5e6 photons per rank instead of 13e6 / 12e6


def test_balanced_split_over_two_gloo_ranks_is_the_one_rank_job(tmp_path):
    """The balanced split keeps batches whole, so two ranks hold exactly the batch moments one rank would: the reduced
    array is BITWISE the sum of the two halves computed in one process, and the straight one-rank sum to rounding (the
    f64 additions associate differently: (a+b+c) + (d+e+f) against a+b+c+d+e+f)."""
    out = str(tmp_path / "reduced.npy")
    mp.spawn(_worker_balanced, args=(2, _free_port(), out), nprocs=2, join=True)
    reduced = np.load(out)
    P = cases.oracle_problem(cases.step_cloud(0.99))
    ppb, nb = driver.balanced_job(PPB * 5, 5, 2)
    halves = [_moments_from_batches(P, (1.0, 0.0), lo, n, 32, 1, 32, ppb=ppb) for lo, n in (driver.split_batches(nb, r, 2) for r in range(2))]
    assert np.array_equal(reduced, halves[0] + halves[1])
    single = _moments_from_batches(P, (1.0, 0.0), 0, nb, 32, 1, 32, ppb=ppb)
    assert reduced[0] == single[0] == ppb * nb and reduced[1] == single[1] == nb
    assert np.allclose(reduced, single, rtol=1e-13, atol=0)


def test_dynamic_schedule_over_two_gloo_ranks(tmp_path):
    """Work units from the shared counter (the reference's master / worker hand-out): every batch once, the faster rank
    takes more of them, and the statistics are the one-rank job's."""
    out = str(tmp_path / "dyn.npy")
    mp.spawn(_worker_dynamic, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    counts, got = got[-2:], got[:-2]
    assert counts.sum() == NB and counts[0] > counts[1] >= 1, counts  # (rank 1 sleeps 0.3 s per unit)
    P = cases.oracle_problem(cases.step_cloud(0.99))
    single = driver.statistics(driver.unpack_moments(_moments_from_batches(P, (1.0, 0.0), 0, NB, 32, 1, 32), 32, 1, 32))
    want = np.concatenate([[single["totalPhotons"], single["batches"], single["meanFluxUp"], single["meanFluxDown"],
                            single["meanFluxAbsorbed"], single["meanFluxUp_StdErr"]], single["absorbedProfile"]])
    assert np.allclose(got, want, rtol=1e-12, atol=1e-15)


def test_two_rank_reduction_equals_single_rank(tmp_path):
    out = str(tmp_path / "reduced.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    reduced = np.load(out)
    case = cases.step_cloud(0.99)
    P = cases.oracle_problem(case)
    single = _moments_from_batches(P, (1.0, 0.0), 0, NB, 32, 1, 32)
    assert np.allclose(reduced, single, rtol=1e-13, atol=0)
    s2 = driver.statistics(driver.unpack_moments(reduced, 32, 1, 32))
    s1 = driver.statistics(driver.unpack_moments(single, 32, 1, 32))
    assert s2["totalPhotons"] == PPB * NB and s2["batches"] == NB
    for k in ("meanFluxUp", "fluxDown", "absorbedProfile", "absorbedVolume_StdErr", "meanFluxAbsorbed_StdErr"):
        assert np.allclose(s2[k], s1[k], rtol=1e-12, atol=1e-15)
    # and the driver statistics are the reference's (monteCarloDriver.f95:1188-1228)
    from oracle import oracle as O
    batches = []
    for b in range(NB):
        r = O.compute_radiative_transfer(P, O.solar_source(1.0, 0.0), O.philox_rng(SEED, b * PPB), PPB)
        batches.append((PPB, np.array([r["meanFluxUp"], r["meanFluxDown"], r["meanFluxAbsorbed"]], np.float64)))
    mean, err = O.batch_statistics(batches)
    assert np.allclose([s1["meanFluxUp"], s1["meanFluxDown"], s1["meanFluxAbsorbed"]], mean, rtol=1e-12)
    assert np.allclose([s1["meanFluxUp_StdErr"], s1["meanFluxDown_StdErr"], s1["meanFluxAbsorbed_StdErr"]], err,
                       rtol=1e-9)
