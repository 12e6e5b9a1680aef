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


def run(integrator, domain, photons, numPhotonsPerBatch, numBatches, randomNumbers, solarFlux=1.0, dist=None,
        moments_tensor=None):
    """Worker loop + reduction.  `dist` is torch.distributed (initialised) or None.
    With dist, `moments_tensor` must be a CUDA double tensor of 8 + 2*M elements that the
    integrator accumulates into (bindMoments) and that is all-reduced in place."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    lo, nb = split_batches(numBatches, rank, world)
    integrator.resetMoments()
    first = randomNumbers.nextPhotonId
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
