"""The part of Drivers/monteCarloDriver.f95 that surrounds the hot path: the worker's
batch loop (:889-1085), the reduction of the batch moments over processes (:1151-1166,
here one RCCL all-reduce over xGMI) and mean / standard error (:1188-1228).

Photon batches shard across ranks with no data-path collective: rank r traces the
batches b with b % world == r ... in contiguous blocks, and because photon ids (not
generator state) drive the random numbers, the union is the same set of photons for
any number of GPUs."""
import numpy as np


def split_batches(numBatches, rank, world):
    """Contiguous block of batches for this rank."""
    base, extra = divmod(numBatches, world)
    lo = rank * base + min(rank, extra)
    return lo, base + (1 if rank < extra else 0)


def balanced_job(totalPhotons, numBatches, world):
    """(photons per batch, batches) of a job of about totalPhotons photons cut so that every rank gets the SAME number of
    whole batches: the smallest multiple of `world` that is >= numBatches.  The reference's master deals work units of
    numPhotonsPerBatch photons to whichever worker is free (monteCarloDriver.f95:665-880), which balances by itself; a
    static split of 100 batches over 8 ranks is 13/13/13/13/12/12/12/12 -- the slowest rank carries 4 % more than the
    mean.  Batches stay whole (they are the statistical units of the standard errors, :1188-1228, and whole batches keep an
    N-rank run bitwise equal to the one-rank run of the same batches), so the batch count moves instead: 100 batches of
    1e6 on 8 ranks become 104 of 961 538 (the job is then 99 999 952 photons; batches are of equal size in the C ABI)."""
    world = max(1, int(world))
    nb = -(-int(numBatches) // world) * world
    ppb = max(1, int(round(float(totalPhotons) / nb)))
    return ppb, nb


def unpack_moments(buf, nx, ny, nz, nDirections=None):
    """header(8) + S1[M] + S2[M] -> dict name -> (S1, S2) in [ix, iy(, iz | direction)] index order."""
    ncol, nvox = nx * ny, nx * ny * nz
    M = 3 + 3 * ncol + nz + nvox
    if nDirections is None:  # the length tells
        nDirections = ((len(buf) - 8) // 2 - M) // ncol
    M += nDirections * ncol
    S1, S2 = buf[8:8 + M], buf[8 + M:8 + 2 * M]
    out = {"totalPhotons": buf[0], "batches": buf[1]}
    names = [("meanFluxUp", 1, None), ("meanFluxDown", 1, None), ("meanFluxAbsorbed", 1, None),
             ("fluxUp", ncol, (ny, nx)), ("fluxDown", ncol, (ny, nx)), ("fluxAbsorbed", ncol, (ny, nx)),
             ("absorbedProfile", nz, None), ("absorbedVolume", nvox, (nz, ny, nx))]
    if nDirections > 0:
        names.append(("intensity", nDirections * ncol, (nDirections, ny, nx)))
    o = 0
    for name, n, shp in names:
        a, b = S1[o:o + n], S2[o:o + n]
        if shp is not None:
            a, b = a.reshape(shp).T, b.reshape(shp).T
        elif n == 1:
            a, b = a[0], b[0]
        out[name] = (a, b)
        o += n
    return out


def statistics(moments, solarFlux=1.0):
    """monteCarloDriver.f95:1188-1228: mean = F*S1/N, stderr = sqrt(max(0, F^2*S2/N - mean^2)/(B-1))."""
    N, B = moments["totalPhotons"], moments["batches"]
    res = {"totalPhotons": int(N), "batches": int(B)}
    for k, v in moments.items():
        if k in ("totalPhotons", "batches"):
            continue
        s1, s2 = v
        mean = solarFlux * s1 / N
        second = solarFlux * (solarFlux * s2 / N)
        res[k] = mean
        res[k + "_StdErr"] = np.sqrt(np.maximum(0.0, second - mean ** 2) / max(B - 1.0, 1.0))
    return res


class UnitCounter:
    """A counter of work-unit ids shared by the ranks of a job: next() hands out 0, 1, 2, ... exactly once each, to
    whichever rank asks first -- the reference's master (monteCarloDriver.f95:665-880 deals work units to the worker that
    is free; worker side :903-918) without a master rank and without its MPI_SEND / MPI_PROBE traffic.  The count lives in
    the process group's key-value store (its `add` is atomic); host side only, nothing crosses the GPUs' links."""
    _sequence = 0  # every rank creates its counters in the same order: the key is the same on all of them

    def __init__(self, dist):
        from torch.distributed import distributed_c10d
        self.store = distributed_c10d._get_default_store()
        UnitCounter._sequence += 1
        self.key = "mcbrat3d_amd/units/%d" % UnitCounter._sequence

    def next(self):
        return int(self.store.add(self.key, 1)) - 1


def run(integrator, domain, photons, numPhotonsPerBatch, numBatches, randomNumbers, solarFlux=1.0, dist=None,
        moments_tensor=None, schedule="static", unitBatches=0):
    """Worker loop + reduction.  `dist` is torch.distributed (initialised) or None.
    With dist, `moments_tensor` must be a CUDA double tensor of 8 + 2*M elements that the
    integrator accumulates into (bindMoments) and that is all-reduced in place.

    schedule = "static" (default): rank r traces its contiguous block of batches in ONE call (split_batches; cut the job
    with balanced_job and every rank's block is the same size).  Which rank traced which batch is fixed, so a run is
    reproducible to the last bit.
    schedule = "dynamic": batches are dealt out in units of `unitBatches` (default: about eight units per rank) from a
    shared counter to whichever rank is free, as the reference's master does -- for jobs whose ranks or units are not
    alike (a busy GPU, wavelengths of unlike optical depth).  Every unit is one synchronous call, so each pays the drain
    of its launch: units should be large next to it.  The same photons are traced whatever the
    assignment; sums over batches then depend on it in the last bits."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    integrator.resetMoments()
    first = randomNumbers.nextPhotonId
    if schedule == "dynamic" and dist is not None and world > 1:
        per = int(unitBatches) if unitBatches and unitBatches > 0 else max(1, int(numBatches) // (8 * world))
        nUnits = -(-int(numBatches) // per)
        counter = UnitCounter(dist)
        photons.numberOfPhotons = max(photons.numberOfPhotons, numBatches * numPhotonsPerBatch)
        while True:
            u = counter.next()
            if u >= nUnits:
                break
            b0 = u * per
            randomNumbers.nextPhotonId = first + b0 * numPhotonsPerBatch
            photons.currentPhoton = 1
            integrator.computeRadiativeTransfer(domain, randomNumbers, photons, numPhotonsPerBatch, min(per, numBatches - b0))
    else:
        if schedule not in ("static", "dynamic"):
            raise ValueError("driver.run: schedule must be 'static' or 'dynamic'")
        lo, nb = split_batches(numBatches, rank, world)
        if nb > 0:
            randomNumbers.nextPhotonId = first + lo * numPhotonsPerBatch
            photons.currentPhoton = 1
            photons.numberOfPhotons = max(photons.numberOfPhotons, nb * numPhotonsPerBatch)
            integrator.computeRadiativeTransfer(domain, randomNumbers, photons, numPhotonsPerBatch, nb)
    randomNumbers.nextPhotonId = first + numBatches * numPhotonsPerBatch
    if dist is not None and world > 1:
        import torch
        if moments_tensor.is_cuda:
            if moments_tensor.device.index != integrator.device:
                raise ValueError("driver.run: moments_tensor lives on cuda:%s, the integrator on cuda:%d"
                                 % (moments_tensor.device.index, integrator.device))
            # The library runs on HIP streams of its own: the all-reduce is ordered after the finish kernels of the
            # calls enqueued so far (also in asynchronous mode, where computeRadiativeTransfer returns before they
            # ran), and the next write to the moments after the all-reduce -- by events, not by a device-wide
            # synchronisation of whichever device happens to be torch's current one.
            stream = torch.cuda.current_stream(moments_tensor.device).cuda_stream
            integrator.streamWaitDone(stream)
            dist.all_reduce(moments_tensor, op=dist.ReduceOp.SUM)  # sumAcrossProcesses, :1151-1166
            integrator.waitStream(stream)
            torch.cuda.synchronize(moments_tensor.device)
        else:  # (gloo rehearsal on CPU tensors: tests/test_multi_rank.py)
            integrator.synchronize()
            dist.all_reduce(moments_tensor, op=dist.ReduceOp.SUM)
        buf = moments_tensor.cpu().numpy()
    else:
        buf = integrator.moments()
    nx, ny, nz = integrator._dims
    return statistics(unpack_moments(buf, nx, ny, nz), solarFlux)
