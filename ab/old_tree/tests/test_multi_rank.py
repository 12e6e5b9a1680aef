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


def _moments_from_batches(P, src_args, first_batch, n_batches, nx, ny, nz):
    """What one rank's device holds after its batches: header + S1 + S2 (driver layout)."""
    from oracle import oracle as O
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
