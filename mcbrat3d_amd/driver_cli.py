"""`python -m mcbrat3d_amd.driver_cli run.nml` -- the reference driver's surface around the HIP
integrator: the five namelists of Drivers/monteCarloDriver.f95:103-121 (same names and defaults,
:58-99), a `.dom` domain file (ncio.read_Domain) or a built-in I3RC generator, batch statistics
(:1188-1228) and the ASCII / NetCDF writers (:1324-1495, :1499-1807).

Monochromatic solar runs (LW_flag < 0).  One process per GPU: start it with
`python -m torch.distributed.run --nproc-per-node N -m mcbrat3d_amd.driver_cli run.nml` to shard the
batches over N GPUs; rank 0 writes the output."""
import os
import re
import sys
import time

import numpy as np

DEFAULTS = {  # monteCarloDriver.f95:58-99
    "radiativetransfer": dict(solarmu=1.0, solarazimuth=0.0, surfacetemp=300.0, lw_flag=-1.0, numlambda=1, calcrayl=True,
                              intensitymus=[], intensityphis=[]),
    "montecarlo": dict(numphotonsperbatch=0, numbatches=100, iseed=10, nphaseintervals=10001),
    "algorithms": dict(useraytracing=True, userussianroulette=True, usehybridphasefunsforintencalcs=False,
                       hybridphasefunwidth=7.0, numordersorigphasefunintencalcs=0, userussianrouletteforintensity=True,
                       zetamin=0.3, limitintensitycontributions=False, maxintensitycontribution=77.0),
    "output": dict(reportvolumeabsorption=False, reportabsorptionprofile=False),
    "filenames": dict(physdomainfile="", domainfilename="", sspfilename="", solarsourcefile="", instrresponsefile="",
                      outputfluxfile="", outputabsproffile="", outputabsvolumefile="", outputnetcdffile="", outputradfile=""),
}


def _value(tok):
    t = tok.strip()
    if not t:
        return None
    if t[0] in "'\"":
        return t[1:-1]
    low = t.lower().strip(".")
    if low in ("true", "t"):
        return True
    if low in ("false", "f"):
        return False
    try:
        return int(t)
    except ValueError:
        return float(t.lower().replace("d", "e"))


def read_namelists(path):
    """Minimal Fortran namelist reader: &group key = value[, value...] ... /"""
    text = open(path).read()
    text = re.sub(r"!.*", "", text)
    groups = {}
    for m in re.finditer(r"&(\w+)(.*?)(?:^|\s)/", text, re.S | re.M):
        body = m.group(2)
        keys = list(re.finditer(r"(\w+)\s*=", body))
        items = {}
        for k, nxt in zip(keys, keys[1:] + [None]):
            raw = body[k.end():nxt.start() if nxt else len(body)]
            vals = [_value(v) for v in re.findall(r"'[^']*'|\"[^\"]*\"|[^,\s]+", raw)]
            vals = [v for v in vals if v is not None]
            items[k.group(1).lower()] = vals[0] if len(vals) == 1 else vals
        groups[m.group(1).lower()] = items
    cfg = {}
    for g, d in DEFAULTS.items():
        cfg.update({k.lower(): v for k, v in d.items()})
        cfg.update(groups.get(g, {}))
    return cfg


def writeResults_ASCII(path, cfg, domainName, stats, xe, ye, ze, solarFlux, albedo):
    """Flux file, monteCarloDriver.f95:1376-1399."""
    b = lambda v: "T" if v else "F"  # noqa: E731
    with open(path, "w") as f:
        f.write("!   I3RC Monte Carlo 3D Solar Radiative Transfer: Flux\n")
        f.write("!  Property_File=%60s\n" % domainName[:60])
        f.write("!  Num_Photons=%10d\n" % stats["totalPhotons"])
        f.write("!  PhotonTracing=%s    Russian_Roulette=%s\n" % (b(cfg["useraytracing"]), b(cfg["userussianroulette"])))
        f.write("!  Hybrid_Phase_Func_for_Radiance=F   Gaussian_Phase_Func_Width_deg= 7.00\n")
        f.write("!  Solar_Flux=%13.6E   Solar_Mu=%10.7f   Solar_Phi=%7.3f\n" % (solarFlux, cfg["solarmu"], cfg["solarazimuth"]))
        f.write("!  Lambertian_Surface_Albedo=%7.4f\n" % albedo)
        f.write("!  Output_Type= Pixel Flux\n")
        f.write("!  Upwelling_Level=%7.3f   Downwelling_level=%7.3f\n" % (ze[-1], ze[0]))
        f.write("!   X      Y           Flux_Up             Flux_Down            Flux_Absorbed \n")
        f.write("!                  Mean     StdErr       Mean     StdErr       Mean     StdErr\n")
        f.write("!  Average:   " + "".join(" %10.4f%10.4f" % (stats[k], stats[k + "_StdErr"])
                                           for k in ("meanFluxUp", "meanFluxDown", "meanFluxAbsorbed")) + "\n")
        for j in range(len(ye) - 1):
            for i in range(len(xe) - 1):
                f.write("%7.3f%7.3f" % (0.5 * (xe[i] + xe[i + 1]), 0.5 * (ye[j] + ye[j + 1])) +
                        "".join(" %10.4f%10.4f" % (stats[k][i, j], stats[k + "_StdErr"][i, j])
                                for k in ("fluxUp", "fluxDown", "fluxAbsorbed")) + "\n")


def writeResults_ASCII_radiance(path, cfg, domainName, stats, xe, ye, ze, solarFlux, albedo, mus, phis):
    """The radiance file, monteCarloDriver.f95:1459-1494."""
    nx, ny, nz = len(xe) - 1, len(ye) - 1, len(ze) - 1
    b = lambda v: "T" if v else "F"  # noqa: E731
    with open(path, "w") as f:
        f.write("!   I3RC Monte Carlo 3D Solar Radiative Transfer: Radiance\n")
        f.write("!  Property_File=%60s\n" % domainName[:60])
        f.write("!  Num_Photons=%10d\n" % int(stats["totalPhotons"]))
        f.write("!  PhotonTracing=%s    Russian_Roulette=%s\n" % (b(cfg["useraytracing"]), b(cfg["userussianroulette"])))
        f.write("!  Hybrid_Phase_Func_for_Radiance=%s   Gaussian_Phase_Func_Width_deg=%5.2f\n"
                % (b(cfg["usehybridphasefunsforintencalcs"]), cfg["hybridphasefunwidth"]))
        f.write("!  Intensity_uses_Russian_Roulette=%s   Intensity_Russian_Roulette_zeta_min=%5.2f\n"
                % (b(cfg["userussianrouletteforintensity"]), cfg["zetamin"]))
        f.write("!  limited_intensity_contributions=%s   max_intensity_contribution=%5.2f\n"
                % (b(cfg["limitintensitycontributions"]), cfg["maxintensitycontribution"]))
        f.write("!  Solar_Flux=%13.6E   Solar_Mu=%10.7f   Solar_Phi=%7.3f\n" % (solarFlux, cfg["solarmu"], cfg["solarazimuth"]))
        f.write("!  Lambertian_Surface_Albedo=%7.4f\n" % albedo)
        f.write("!  Output_Type= Pixel Radiance\n")
        f.write("!  RADIANCE AT Z=%7.3f   NXO=%4d   NYO=%4d   NDIR=%4d\n" % (ze[nz], nx, ny, len(mus)))
        f.write("!   X      Y         Radiance (Mean, StdErr)\n")
        for k in range(len(mus)):
            f.write("!  %8.5f %6.2f  <- (mu,phi)\n" % (mus[k], phis[k]))
            for j in range(ny):
                for i in range(nx):
                    f.write("%7.3f%7.3f %9.4f %9.4f\n" % (0.5 * (xe[i] + xe[i + 1]), 0.5 * (ye[j] + ye[j + 1]),
                                                        stats["intensity"][i, j, k], stats["intensity_StdErr"][i, j, k]))


def builtin_domain(name):
    """Domain-Files/i3rcStepCloud.f95, planeParallel.f95 (km)."""
    import mcbrat3d_amd as M
    g = float(np.float32(0.85))
    coef = np.array([g ** l for l in range(1, 65)]).astype(np.float32)
    table = M.new_PhaseFunctionTable([M.new_PhaseFunction(coef)])
    ssa = 1.0 if "Conservative" in name else 0.99
    if name.startswith("i3rcStepCloud"):
        nx = 32
        xe = 0.015625 * np.arange(nx + 1)
        ext = np.zeros((nx, 1, 32)); ext[:16] = 8.0; ext[16:] = 72.0
    elif name.startswith("planeParallel"):
        xe = np.array([0.0, 0.5])
        ext = np.full((1, 1, 32), 2.0)
    else:
        raise SystemExit("unknown builtin domain " + name)
    dom = M.new_Domain(xe, [0.0, 0.5], 0.0078125 * np.arange(33), surfaceAlbedo=0.0)
    dom.addOpticalComponent("cloud", ext, np.full_like(ext, ssa), np.ones(ext.shape, np.int32), table)
    return dom


def load_domains(cfg):
    """The optical domain of every wavelength of the run.  The current driver's file family (monteCarloDriver.f95:299,
    :936): `physDomainFile` (read_Common) + up to four `SSPfilename` tables (read_SSPTable), numLambda wavelengths,
    Rayleigh scattering added when calcRayl.  Otherwise one domain: a `.dom` file (read_Domain) or a built-in I3RC
    generator ("builtin:i3rcStepCloud")."""
    from mcbrat3d_amd import ncio
    ssp = cfg["sspfilename"]
    ssp = [ssp] if isinstance(ssp, str) else list(ssp)
    ssp = [n for n in ssp if n]
    if ssp:
        if not cfg["physdomainfile"]:
            raise SystemExit("SSPfilename given without physDomainFile")
        common = ncio.read_Common(cfg["physdomainfile"])
        return [ncio.read_SSPTable(ssp, i, common, setup=False, calcRayl=bool(cfg["calcrayl"]))
                for i in range(1, int(cfg["numlambda"]) + 1)]
    domfile = cfg["physdomainfile"] or cfg["domainfilename"]
    dom = builtin_domain(domfile[8:]) if domfile.startswith("builtin:") else ncio.read_Domain(domfile)
    dom.getOpticalPropertiesByComponent()
    return [dom]


def run_spectral(cfg, doms, rank, world, local, dist):
    """Spectrally integrated run (numLambda > 1, or thermal emission): every wavelength resident on the GPU
    (broadband.SpectralRun), photons split over wavelengths on the device, ranks take contiguous blocks of batches."""
    import mcbrat3d_amd as M
    from mcbrat3d_amd import broadband, driver, ncio
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    run = broadband.SpectralRun(M, doms, device=local, minInverseTableSize=cfg["nphaseintervals"],
                                useRayTracing=cfg["useraytracing"], useRussianRoulette=cfg["userussianroulette"])
    if cfg["lw_flag"] >= 0:
        flux = run.prepare_thermal(cfg["surfacetemp"])
    else:
        nlam = len(doms)
        if not cfg["solarsourcefile"]:
            raise SystemExit("a solar run over several wavelengths needs solarSourceFile")
        src, lam = ncio.read_SolarSource(cfg["solarsourcefile"], nlam)
        srf = ncio.read_specResponseFunction(cfg["instrresponsefile"], nlam) if cfg["instrresponsefile"] else None
        flux = run.prepare_solar(cfg["solarmu"], cfg["solarazimuth"], src, lam, srf)
    ppb, nbAll = int(cfg["numphotonsperbatch"]), int(cfg["numbatches"])
    lo, nb = driver.split_batches(nbAll, rank, world)
    moments = None
    if dist is not None:
        import torch
        moments = torch.zeros(8 + 2 * run.first.momentsLength(), dtype=torch.float64, device="cuda:%d" % local)
        run.bindMoments(moments.data_ptr())
    run.resetMoments()
    if nb > 0:
        # this rank's photons: ids and wavelength draws [lo * ppb, (lo + nb) * ppb)
        counts = broadband.device_frequency_distribution(run.first, run.cdf, ppb * nb, cfg["iseed"], firstDraw=lo * ppb)
        run.run(ppb, nb, new_RandomNumberSequence(cfg["iseed"], lo * ppb), seed=cfg["iseed"], counts=counts)
    for it in run.integrators:
        it.synchronize()
    if dist is not None and world > 1:
        import torch
        dist.all_reduce(moments, op=dist.ReduceOp.SUM)  # sumAcrossProcesses, monteCarloDriver.f95:1151-1166
        torch.cuda.synchronize()
        buf = moments.cpu().numpy()
    else:
        buf = run.moments()
    nx, ny, nz = run.first._dims
    stats = driver.statistics(driver.unpack_moments(buf, nx, ny, nz), solarFlux=flux)
    run.finalize()
    return stats, flux


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        raise SystemExit("usage: python -m mcbrat3d_amd.driver_cli <namelist file>")
    t0 = time.time()
    cfg = read_namelists(argv[0])
    if cfg["numphotonsperbatch"] <= 0:
        raise SystemExit("must specify numPhotonsPerBatch")
    import mcbrat3d_amd as M
    from mcbrat3d_amd import driver, ncio
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    domfile = cfg["physdomainfile"] or cfg["domainfilename"]
    doms = load_domains(cfg)
    dom = doms[0]
    if len(doms) > 1 or cfg["lw_flag"] >= 0:
        setup = time.time() - t0
        stats, flux = run_spectral(cfg, doms, rank, world, local, dist)
        if rank == 0:
            print(" spectrally integrated flux %.6g; mean flux up/down/absorbed: " % flux +
                  "  ".join("%9.6f +-%9.6f" % (stats[k], stats[k + "_StdErr"]) for k in ("meanFluxUp", "meanFluxDown", "meanFluxAbsorbed")))
            xe, ye, ze = dom.xPosition, dom.yPosition, dom.zPosition
            if cfg["outputfluxfile"]:
                writeResults_ASCII(cfg["outputfluxfile"], cfg, domfile, stats, xe, ye, ze, flux, dom.surfaceAlbedo)
            if cfg["outputnetcdffile"]:
                ncio.writeResults_netcdf(cfg["outputnetcdffile"], domfile, stats, xe, ye, ze, solarFlux=flux, solarMu=cfg["solarmu"],
                                         solarAzimuth=cfg["solarazimuth"], surfaceAlbedo=dom.surfaceAlbedo, iseed=cfg["iseed"],
                                         nPhaseIntervals=cfg["nphaseintervals"], reportAbsorptionProfile=cfg["reportabsorptionprofile"],
                                         reportVolumeAbsorption=cfg["reportvolumeabsorption"], cpuTimeTotal=time.time() - t0,
                                         cpuTimeSetup=setup, numProcs=world)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return stats
    integ = M.new_Integrator(dom, device=local)
    integ.specifyParameters(minInverseTableSize=cfg["nphaseintervals"], useRayTracing=cfg["useraytracing"],
                            useRussianRoulette=cfg["userussianroulette"], LW_flag=cfg["lw_flag"])
    # intensity directions: every entry with |mu| > 0, and only if a file will hold them (monteCarloDriver.f95:279-282)
    mus = np.atleast_1d(np.asarray(cfg["intensitymus"], np.float32))
    phis = np.atleast_1d(np.asarray(cfg["intensityphis"], np.float32))
    if phis.size < mus.size:
        phis = np.concatenate([phis, np.zeros(mus.size - phis.size, np.float32)])
    keep = np.abs(mus) > 0.0
    mus, phis = mus[keep], phis[:mus.size][keep]
    computeIntensity = mus.size > 0 and bool(cfg["outputradfile"] or cfg["outputnetcdffile"])
    if computeIntensity:  # :547-552, :579-594
        integ.specifyParameters(minForwardTableSize=cfg["nphaseintervals"], intensityMus=mus, intensityPhis=phis,
                                computeIntensity=True,
                                useHybridPhaseFunsForIntenCalcs=cfg["usehybridphasefunsforintencalcs"],
                                hybridPhaseFunWidth=cfg["hybridphasefunwidth"],
                                numOrdersOrigPhaseFunIntenCalcs=cfg["numordersorigphasefunintencalcs"],
                                useRussianRouletteForIntensity=cfg["userussianrouletteforintensity"], zetaMin=cfg["zetamin"],
                                limitIntensityContributions=cfg["limitintensitycontributions"],
                                maxIntensityContribution=cfg["maxintensitycontribution"])
    photons = M.new_PhotonStream(cfg["solarmu"], cfg["solarazimuth"], numberOfPhotons=cfg["numphotonsperbatch"] * cfg["numbatches"])
    moments = None
    if dist is not None:
        import torch
        moments = torch.zeros(8 + 2 * integ.momentsLength(), dtype=torch.float64, device="cuda:%d" % local)
        integ.bindMoments(moments.data_ptr())
    setup = time.time() - t0
    stats = driver.run(integ, dom, photons, cfg["numphotonsperbatch"], cfg["numbatches"], new_RandomNumberSequence(cfg["iseed"]),
                       solarFlux=1.0, dist=dist, moments_tensor=moments)
    if rank == 0:
        print(" mean flux up/down/absorbed: " + "  ".join("%9.6f +-%9.6f" % (stats[k], stats[k + "_StdErr"])
                                                            for k in ("meanFluxUp", "meanFluxDown", "meanFluxAbsorbed")))
        xe, ye, ze = dom.xPosition, dom.yPosition, dom.zPosition
        if cfg["outputfluxfile"]:
            writeResults_ASCII(cfg["outputfluxfile"], cfg, domfile, stats, xe, ye, ze, 1.0, dom.surfaceAlbedo)
        if cfg["outputradfile"] and computeIntensity:
            writeResults_ASCII_radiance(cfg["outputradfile"], cfg, domfile, stats, xe, ye, ze, 1.0, dom.surfaceAlbedo, mus, phis)
        if cfg["outputnetcdffile"]:
            ncio.writeResults_netcdf(cfg["outputnetcdffile"], domfile, stats, xe, ye, ze, solarFlux=1.0, solarMu=cfg["solarmu"],
                                     solarAzimuth=cfg["solarazimuth"], surfaceAlbedo=dom.surfaceAlbedo, iseed=cfg["iseed"],
                                     nPhaseIntervals=cfg["nphaseintervals"],
                                     reportAbsorptionProfile=cfg["reportabsorptionprofile"],
                                     reportVolumeAbsorption=cfg["reportvolumeabsorption"],
                                     cpuTimeTotal=time.time() - t0, cpuTimeSetup=setup, numProcs=world,
                                     intensityMus=mus if computeIntensity else None,
                                     intensityPhis=phis if computeIntensity else None,
                                     useHybridPhaseFunsForIntenCalcs=cfg["usehybridphasefunsforintencalcs"],
                                     hybridPhaseFunWidth=cfg["hybridphasefunwidth"],
                                     useRussianRouletteForIntensity=cfg["userussianrouletteforintensity"], zetaMin=cfg["zetamin"],
                                     limitIntensityContributions=cfg["limitintensitycontributions"],
                                     maxIntensityContribution=cfg["maxintensitycontribution"])
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    integ.finalize()
    return stats


if __name__ == "__main__":
    main()
