"""NetCDF-3 files in the reference's layouts, read and written with scipy.io.netcdf_file (this image
has no netCDF library; the formats are plain NetCDF classic / 64-bit offset, which scipy handles).

* write_Domain / read_Domain  -- the `.dom` optical-domain file of src/opticalProperties.f95:1087-1427
  with the per-component phase-function tables of src/scatteringPhaseFunctions.f95:931-1118
  (add_PhaseFunctionTable) / :1279-1440.
* writeResults_netcdf         -- the result file of Drivers/monteCarloDriver.f95:1499-1807.

Fortran writes arrays dimensioned (x, y, z); NetCDF stores dimensions slowest-first, so the same
variable appears here with shape (z, y, x)."""
import numpy as np
from scipy.io import netcdf_file

from ._capi import McbratError
from .domain import Domain
from .phase import PhaseFunction, PhaseFunctionTable


def _prefix(i):
    return "Component%d_" % i  # makePrefix, opticalProperties.f95:1611-1621


def write_Domain(thisDomain, fileName):
    info = thisDomain.getInfo_Domain()
    nx, ny, nz = info["numX"], info["numY"], info["numZ"]
    f = netcdf_file(fileName, "w", version=1)  # nf90_Clobber: classic format (:1108)
    try:
        for name, n in (("x-Edges", nx + 1), ("y-Edges", ny + 1), ("z-Edges", nz + 1), ("x-Grid", nx), ("y-Grid", ny), ("z-Grid", nz)):
            f.createDimension(name, n)
        for name, arr in (("x-Edges", info["xPosition"]), ("y-Edges", info["yPosition"]), ("z-Edges", info["zPosition"])):
            f.createVariable(name, "d", (name,))[:] = arr
        temps = np.zeros((nx, ny, nz)) if thisDomain.temps is None else thisDomain.temps
        f.createVariable("Temperatures", "d", ("z-Grid", "y-Grid", "x-Grid"))[:] = temps.transpose(2, 1, 0)
        from .integrator import Integrator  # regular-spacing flags as new_Domain computes them (:496-503)
        dx, dy, dz = (np.diff(info[k]) for k in ("xPosition", "yPosition", "zPosition"))
        xyreg = bool(np.all(np.abs(dx - dx[0]) <= 2 * np.spacing(info["xPosition"][1:])) and
                     np.all(np.abs(dy - dy[0]) <= 2 * np.spacing(info["yPosition"][1:])))
        zreg = bool(np.all(np.abs(dz - dz[0]) <= 2 * np.spacing(info["zPosition"][1:])))
        f.xyRegularlySpaced = np.int8(xyreg)
        f.zRegularlySpaced = np.int8(zreg)
        f.__setattr__("lambda", np.float64(thisDomain.lambda_um))
        f.lambdaIndex = np.int32(1)
        f.numberOfLambdas = np.int32(1)
        f.surfaceAlbedo = np.float64(thisDomain.surfaceAlbedo)
        f.numberOfComponents = np.int32(len(thisDomain.components))
        for i, comp in enumerate(thisDomain.components, start=1):
            p = _prefix(i)
            f.__setattr__(p + "Name", comp["name"])
            f.__setattr__(p + "zLevelBase", np.int32(comp["zLevelBase"]))
            nzc = comp["ext"].shape[-1]
            zdim = "z-Grid"
            if not (comp["zLevelBase"] == 1 and nzc == nz):  # fillsDomainInVertical (:1146-1153)
                zdim = p + "z-Grid"
                f.createDimension(zdim, nzc)
            if comp["ext"].ndim == 1:  # horizontally uniform (:1154-1163)
                dims = (zdim,)
                tr = lambda a: a  # noqa: E731
            else:
                dims = (zdim, "y-Grid", "x-Grid")
                tr = lambda a: a.transpose(2, 1, 0)  # noqa: E731
            f.createVariable(p + "Extinction", "d", dims)[:] = tr(comp["ext"])
            f.createVariable(p + "SingleScatteringAlbedo", "d", dims)[:] = tr(comp["ssa"])
            f.createVariable(p + "PhaseFunctionIndex", "h", dims)[:] = tr(comp["pfIndex"]).astype(np.int16)
            _add_PhaseFunctionTable(f, comp["table"], p)
    finally:
        f.close()
    return fileName


def _add_PhaseFunctionTable(f, table, p):
    """scatteringPhaseFunctions.f95:1005-1105."""
    n = table.nEntries
    f.createDimension(p + "phaseFunctionNumber", n)
    f.createVariable(p + "phaseFunctionKeyT", "f", (p + "phaseFunctionNumber",))[:] = table.key
    f.createVariable(p + "extinctionT", "d", (p + "phaseFunctionNumber",))[:] = [q.extinction for q in table.phaseFunctions]
    f.createVariable(p + "singleScatteringAlbedoT", "d", (p + "phaseFunctionNumber",))[:] = \
        [q.singleScatteringAlbedo for q in table.phaseFunctions]
    if table.description:
        f.__setattr__(p + "description", table.description)
    legendre = [q.legendreCoefficients is not None for q in table.phaseFunctions]
    if all(legendre):
        length = np.array([len(q.legendreCoefficients) for q in table.phaseFunctions], np.int32)
        start = np.concatenate([[1], 1 + np.cumsum(length[:-1])]).astype(np.int32)
        f.createDimension(p + "coefficents", int(max(start[-1] + length[-1] - 1, 1)))  # (sic) :1046
        f.createVariable(p + "start", "i", (p + "phaseFunctionNumber",))[:] = start
        f.createVariable(p + "length", "i", (p + "phaseFunctionNumber",))[:] = length
        coeffs = np.concatenate([q.legendreCoefficients for q in table.phaseFunctions]) if length.sum() else np.zeros(1, np.float32)
        f.createVariable(p + "legendreCoefficients", "f", (p + "coefficents",))[:] = coeffs
        f.__setattr__(p + "phaseFunctionStorageType", "LegendreCoefficients")
    elif not any(legendre):
        ang = table.phaseFunctions[0].scatteringAngle
        if any(len(q.scatteringAngle) != len(ang) or np.any(q.scatteringAngle != ang) for q in table.phaseFunctions):
            raise McbratError("add_PhaseFunctionTable: angle-value tables must share one set of angles")
        f.createDimension(p + "scatteringAngle", len(ang))
        f.createVariable(p + "scatteringAngle", "f", (p + "scatteringAngle",))[:] = ang
        f.createVariable(p + "phaseFunctionValues", "f", (p + "phaseFunctionNumber", p + "scatteringAngle"))[:] = \
            np.stack([q.value for q in table.phaseFunctions])
        f.__setattr__(p + "phaseFunctionStorageType", "Angle-Value")
    else:
        raise McbratError("add_PhaseFunctionTable: mixed storage types in one table")


def _read_PhaseFunctionTable(f, p):
    att = f.__dict__ if hasattr(f, "__dict__") else {}
    storage = getattr(f, p + "phaseFunctionStorageType")
    storage = storage.decode() if isinstance(storage, bytes) else storage
    key = np.array(f.variables[p + "phaseFunctionKeyT"][:], np.float32)
    ext = np.array(f.variables[p + "extinctionT"][:])
    ssa = np.array(f.variables[p + "singleScatteringAlbedoT"][:])
    pfs = []
    if storage == "LegendreCoefficients":
        start = np.array(f.variables[p + "start"][:]); length = np.array(f.variables[p + "length"][:])
        coeffs = np.array(f.variables[p + "legendreCoefficients"][:], np.float32)
        for s, n, e, w in zip(start, length, ext, ssa):
            pfs.append(PhaseFunction(legendreCoefficients=coeffs[s - 1:s - 1 + n].copy(), extinction=float(e), singleScatteringAlbedo=float(w)))
    elif storage == "Angle-Value":
        ang = np.array(f.variables[p + "scatteringAngle"][:], np.float32)
        vals = np.array(f.variables[p + "phaseFunctionValues"][:], np.float32)
        for v, e, w in zip(vals, ext, ssa):
            pfs.append(PhaseFunction(scatteringAngle=ang.copy(), value=v.copy(), extinction=float(e), singleScatteringAlbedo=float(w)))
    else:
        raise McbratError("read_PhaseFunctionTable: unknown phaseFunctionStorageType " + str(storage))
    desc = getattr(f, p + "description", b"")
    return PhaseFunctionTable(pfs, key, desc.decode() if isinstance(desc, bytes) else desc)


def read_Domain(fileName):
    """read_Domain (src/opticalProperties.f95:1253-1427)."""
    f = netcdf_file(fileName, "r", mmap=False)
    try:
        dom = Domain(np.array(f.variables["x-Edges"][:]), np.array(f.variables["y-Edges"][:]), np.array(f.variables["z-Edges"][:]),
                     temps=np.array(f.variables["Temperatures"][:]).transpose(2, 1, 0) if "Temperatures" in f.variables else None,
                     surfaceAlbedo=float(getattr(f, "surfaceAlbedo", 0.0)), lambda_um=float(getattr(f, "lambda", 0.0)))
        for i in range(1, int(getattr(f, "numberOfComponents", 0)) + 1):
            p = _prefix(i)
            name = getattr(f, p + "Name")
            fix = lambda a: a.transpose(2, 1, 0) if a.ndim == 3 else a  # noqa: E731
            dom.addOpticalComponent(name.decode() if isinstance(name, bytes) else name,
                                    fix(np.array(f.variables[p + "Extinction"][:])),
                                    fix(np.array(f.variables[p + "SingleScatteringAlbedo"][:])),
                                    fix(np.array(f.variables[p + "PhaseFunctionIndex"][:]).astype(np.int32)),
                                    _read_PhaseFunctionTable(f, p), zLevelBase=int(getattr(f, p + "zLevelBase")))
    finally:
        f.close()
    return dom


def writeResults_netcdf(outputFileName, domainFileName, stats, xPosition, yPosition, zPosition, solarFlux=1.0, solarMu=1.0,
                        solarAzimuth=0.0, surfaceAlbedo=0.0, iseed=10, nPhaseIntervals=10001, useRayTracing=True,
                        reportAbsorptionProfile=False, reportVolumeAbsorption=False, cpuTimeTotal=0.0, cpuTimeSetup=0.0,
                        numProcs=1, intensityMus=None, intensityPhis=None, useHybridPhaseFunsForIntenCalcs=False,
                        hybridPhaseFunWidth=0.0, useRussianRouletteForIntensity=False, zetaMin=0.0,
                        limitIntensityContributions=False, maxIntensityContribution=0.0):
    """monteCarloDriver.f95:1499-1807.  `stats` is driver.statistics(...) output ([ix, iy(, iz)] arrays)."""
    xe, ye, ze = (np.asarray(a, np.float64) for a in (xPosition, yPosition, zPosition))
    f = netcdf_file(outputFileName, "w", version=2)  # nf90_64bit_offset (:1559)
    try:
        f.description = "Output from I3RC Community Monte Carlo Model"
        f.Domain_filename = domainFileName
        f.Surface_albedo = np.float64(surfaceAlbedo)
        f.Total_number_of_photons = np.int32(min(int(stats["totalPhotons"]), 2 ** 31 - 1))  # NetCDF-3 has no 64-bit integers
        f.Number_of_batches = np.int32(stats["batches"])
        f.Solar_flux = np.float64(solarFlux)
        f.Solar_mu = np.float32(solarMu)
        f.Solar_phi = np.float32(solarAzimuth)
        f.Random_number_seed = np.int32(iseed)
        f.Phase_function_table_sizes = np.int32(nPhaseIntervals)
        f.Algorithm = "Ray_tracing" if useRayTracing else "Max_cross_section"
        f.Intensity_uses_hyrbid_phase_functions = np.int32(1 if useHybridPhaseFunsForIntenCalcs else 0)  # (sic) :1582-1589
        f.Hybrid_phase_function_width = np.float32(hybridPhaseFunWidth if useHybridPhaseFunsForIntenCalcs else 0.0)
        f.Intensity_uses_Russian_roulette = np.int32(1 if useRussianRouletteForIntensity else 0)
        f.Intensity_Russian_roulette_zeta_min = np.float32(zetaMin if useRussianRouletteForIntensity else 0.0)
        f.limited_intensity_contributions = np.int32(1 if limitIntensityContributions else 0)
        f.max_intensity_contribution = np.float32(maxIntensityContribution if limitIntensityContributions else 0.0)
        f.Cpu_time_total = np.float32(cpuTimeTotal)
        f.Cpu_time_setup = np.float32(cpuTimeSetup)
        f.Number_of_processors_used = np.int32(numProcs)
        f.createDimension("x", len(xe) - 1)
        f.createDimension("y", len(ye) - 1)
        withZ = reportAbsorptionProfile or reportVolumeAbsorption
        if withZ:
            f.createDimension("z", len(ze) - 1)
        f.createVariable("x", "d", ("x",))[:] = 0.5 * (xe[1:] + xe[:-1])  # cell mid-points (:1714-1722)
        f.createVariable("y", "d", ("y",))[:] = 0.5 * (ye[1:] + ye[:-1])
        if withZ:
            f.createVariable("z", "d", ("z",))[:] = 0.5 * (ze[1:] + ze[:-1])
        for name in ("fluxUp", "fluxDown", "fluxAbsorbed"):
            f.createVariable(name, "f", ("y", "x"))[:] = np.asarray(stats[name]).T
            f.createVariable(name + "_StdErr", "f", ("y", "x"))[:] = np.asarray(stats[name + "_StdErr"]).T
        if reportAbsorptionProfile:
            f.createVariable("absorptionProfile", "f", ("z",))[:] = stats["absorbedProfile"]
            f.createVariable("absorptionProfile_StdErr", "f", ("z",))[:] = stats["absorbedProfile_StdErr"]
        if reportVolumeAbsorption:
            f.createVariable("absorbedVolume", "f", ("z", "y", "x"))[:] = np.asarray(stats["absorbedVolume"]).transpose(2, 1, 0)
            f.createVariable("absorbedVolume_StdErr", "f", ("z", "y", "x"))[:] = np.asarray(stats["absorbedVolume_StdErr"]).transpose(2, 1, 0)
        if intensityMus is not None and "intensity" in stats:  # :1666-1676, :1767-1778 (Fortran dims x, y, direction)
            f.createDimension("direction", len(intensityMus))
            f.createVariable("intensityMus", "f", ("direction",))[:] = np.asarray(intensityMus, np.float32)
            f.createVariable("intensityPhis", "f", ("direction",))[:] = np.asarray(intensityPhis, np.float32)
            f.createVariable("intensity", "f", ("direction", "y", "x"))[:] = np.asarray(stats["intensity"]).transpose(2, 1, 0)
            f.createVariable("intensity_StdErr", "f", ("direction", "y", "x"))[:] = np.asarray(stats["intensity_StdErr"]).transpose(2, 1, 0)
    finally:
        f.close()
    return outputFileName


# ------------------------------------------------------------------------------------------------------------
# The file family the current driver reads (Drivers/monteCarloDriver.f95:299, :936): a physical-properties file
# (read_Common) plus up to four single-scattering-property table files (read_SSPTable).  Names are spelled as
# those readers spell them -- lower-case "x-edges", "ExtinctionT" -- which is NOT how write_Domain spells them
# ("x-Edges", "extinctionT"): the two families do not read each other's files.
# ------------------------------------------------------------------------------------------------------------
LIGHT_SPD = 2.99792458E8   # [m/s]   opticalProperties.f95:27-29
AVOGADRO = 6.02214129E23   # [mol^-1]
RSTAR = 8.3144621          # [J K^-1 mol^-1]


class CommonDomain:
    """type(commonDomain), opticalProperties.f95:63-75: what all wavelengths share."""

    def __init__(self, xPosition, yPosition, zPosition, temps, numConc=None, massConc=None, Reff=None, rho=None):
        self.xPosition = np.ascontiguousarray(xPosition, np.float64)
        self.yPosition = np.ascontiguousarray(yPosition, np.float64)
        self.zPosition = np.ascontiguousarray(zPosition, np.float64)
        self.temps, self.numConc, self.massConc, self.Reff, self.rho = temps, numConc, massConc, Reff, rho


def _xyz(a):
    """NetCDF (z, y, x) -> Fortran index order [ix, iy, iz]."""
    return np.array(a, np.float64).transpose(2, 1, 0)


def read_Common(fileName):
    """read_Common, src/opticalProperties.f95:347-451."""
    try:
        f = netcdf_file(fileName, "r", mmap=False)
    except (OSError, TypeError, ValueError):
        raise McbratError("read_Common: Can't open file " + str(fileName))
    try:
        for d in ("x-edges", "y-edges", "z-edges", "z-grid"):
            if d not in f.dimensions:
                raise McbratError("read_Common: %s problem reading dimensions." % fileName)
        for v in ("x-edges", "y-edges", "z-edges", "Temperatures"):
            if v not in f.variables:
                raise McbratError("read_Common: %s doesn't look an optical properties file." % fileName)
        xe, ye, ze = (np.array(f.variables[k][:], np.float64) for k in ("x-edges", "y-edges", "z-edges"))
        nx, ny, nz = len(xe) - 1, len(ye) - 1, len(ze) - 1
        c = CommonDomain(xe, ye, ze, _xyz(f.variables["Temperatures"][:]))

        def spread(var, what):  # 3-D as stored, or a profile copied to every column (:401-410, :436-444)
            a = np.array(var[:], np.float64)
            if a.ndim == 3:
                return a.transpose(2, 1, 0)
            if a.ndim == 1:
                return np.broadcast_to(a[None, None, :], (nx, ny, nz)).copy()
            raise McbratError("read_Common: %s strange number of dimensions for %s" % (fileName, what))

        if "Pressures" in f.variables:  # hPa -> molecules m^-3 (:424)
            prssr = spread(f.variables["Pressures"], "pressure")
            c.numConc = (prssr * 100.0 * AVOGADRO) / (RSTAR * c.temps)
            if "nonGasComps" in f.dimensions and f.dimensions["nonGasComps"]:
                if "massConc" not in f.variables or "Reff" not in f.variables:
                    raise McbratError("read_Common: %s problem reading DENSITY, massConc, or Reff." % fileName)
                # Fortran (component, x, y, z) = NetCDF (z, y, x, component)
                c.massConc = np.array(f.variables["massConc"][:], np.float64).transpose(3, 2, 1, 0)
                c.Reff = np.array(f.variables["Reff"][:], np.float64).transpose(3, 2, 1, 0)
        if "Density" in f.variables:
            c.rho = spread(f.variables["Density"], "density")
        return c
    finally:
        f.close()


def calc_RayleighScattering(lambda_um, rho, N):
    """calc_RayleighScattering, src/opticalProperties.f95:2052-2086: (extinction [km^-1], ssa, phase index, table) of the
    molecular atmosphere from the density and number-concentration profiles."""
    Pi = float(np.float32(3.14159265358979312))  # (the module's Pi is a default real, :26)
    f, rho0 = 1.060816681, 1.275
    lam = float(lambda_um)
    mr1 = 6.4328E-5 + (2.94981E-2 / (146 - (lam ** (-2)))) + (2.554E-4 / (41 - (lam ** (-2))))
    rho, N = np.asarray(rho, np.float64), np.asarray(N, np.float64)
    ext = (32.0E27) * f * (Pi ** 3) * (rho ** 2) * (mr1 ** 2) / (3.0 * N * (rho0 ** 2) * (lam ** 4))
    LG = (np.array([0.0, 0.5], np.float32) / np.array([2.0 * 1.0 + 1.0, 2.0 * 2.0 + 1.0], np.float32)).astype(np.float32)
    table = PhaseFunctionTable([PhaseFunction(legendreCoefficients=LG)], np.zeros(1, np.float32), "Rayleigh Scattering")
    return ext, np.ones_like(ext), np.ones(ext.shape, np.int32), table


def _read_PhaseFunctionTableNEW(f, p, spectIndex):
    """read_PhaseFunctionTableNEW, src/scatteringPhaseFunctions.f95:1279-1440: the table of ONE spectral index out of a
    file whose extinctionT / ssaT / start / length / legendreCoefficients carry a spectral dimension."""
    att = p + "phaseFunctionStorageType"
    if not hasattr(f, att):
        raise McbratError("read_PhaseFunctionTable: file doesn't contain this phase function.")
    storage = getattr(f, att)
    storage = storage.decode() if isinstance(storage, bytes) else storage
    k = spectIndex - 1
    key = np.array(f.variables[p + "phaseFunctionKeyT"][:], np.float32)
    ext = np.array(f.variables[p + "ExtinctionT"][:], np.float64)[k]                  # Fortran (entry, lambda)
    ssa = np.array(f.variables[p + "SingleScatteringAlbedoT"][:], np.float64)[k]
    desc = getattr(f, p + "description", b"")
    desc = desc.decode() if isinstance(desc, bytes) else desc
    pfs = []
    if storage.startswith("Angle-Value"):
        ang = np.array(f.variables[p + "scatteringAngle"][:], np.float32)
        vals = np.array(f.variables[p + "phaseFunctionValues"][:], np.float32)      # Fortran (angle, entry)
        for v, e, w in zip(vals, ext, ssa):
            pfs.append(PhaseFunction(scatteringAngle=ang.copy(), value=v.copy(), extinction=float(e), singleScatteringAlbedo=float(w)))
    elif storage.startswith("LegendreCoefficients"):
        start = np.array(f.variables[p + "start"][:], np.int64)[k]
        length = np.array(f.variables[p + "length"][:], np.int64)[k]
        coeffs = np.array(f.variables[p + "legendreCoefficients"][:], np.float32)[k]  # Fortran (coefficient, lambda)
        first = int(start[0])  # the coefficients of this index are read from start(1) on, starts renumbered from 1 (:1379-1382)
        block = coeffs[first - 1:first - 1 + int(length.sum())]
        start = start - first + 1
        for s, n, e, w in zip(start, length, ext, ssa):
            pfs.append(PhaseFunction(legendreCoefficients=block[s - 1:s - 1 + n].copy(), extinction=float(e), singleScatteringAlbedo=float(w)))
    else:
        raise McbratError("read_PhaseFunctionTable: file is of unknown format.")
    return PhaseFunctionTable(pfs, key, desc)


def read_SSPTable(fileNames, lambdaIndex, commonD, setup=False, calcRayl=True):
    """read_SSPTable, src/opticalProperties.f95:147-345: the optical domain of wavelength `lambdaIndex` (1-based) from
    the common physical domain and up to four single-scattering-property table files.  Gaseous components
    (`extType = "absXsec"`): absorption cross-section profile x number concentration; condensed components
    (`"volExt"`): mass concentration x the table's extinction interpolated in effective radius, the phase function of
    the nearer table entry.  `setup`: only what emission weighting needs (no phase functions, no Rayleigh)."""
    if isinstance(fileNames, str):
        fileNames = [fileNames]
    nx, ny, nz = len(commonD.xPosition) - 1, len(commonD.yPosition) - 1, len(commonD.zPosition) - 1
    dom = None
    comp, gasComp = 1, 0  # counters over all files (:177-178)
    for fileName in [n for n in fileNames if n]:
        try:
            f = netcdf_file(fileName, "r", mmap=False)
        except (OSError, TypeError, ValueError):
            raise McbratError("read_SSPTable: Can't open file " + str(fileName))
        try:
            if "f_grid_nelem" not in f.dimensions or "f_grid" not in f.variables or "surfaceAlbedo" not in f.variables:
                raise McbratError("read_SSPTable: doesn't look an optical properties file.")
            nLambda = int(f.dimensions["f_grid_nelem"])
            if not (1 <= lambdaIndex <= nLambda):
                raise McbratError("read_SSPTable: Error reading scalar fields from file")
            freq = float(np.array(f.variables["f_grid"][:], np.float64)[lambdaIndex - 1])
            lam = (LIGHT_SPD * (10 ** 6)) / freq  # [microns] :194
            albedo = float(np.array(f.variables["surfaceAlbedo"][:], np.float64)[lambdaIndex - 1])
            if dom is None:  # new_DomainBB(commonD, lambda, lambdaIndex, nlambda, albedo) :203
                dom = Domain(commonD.xPosition, commonD.yPosition, commonD.zPosition, temps=commonD.temps,
                             surfaceAlbedo=albedo, lambda_um=lam)
                dom.lambdaIndex, dom.numberOfLambdas = int(lambdaIndex), nLambda
            for i in range(1, int(getattr(f, "numberOfComponents")) + 1):
                p = _prefix(i)
                name = getattr(f, p + "Name")
                name = name.decode() if isinstance(name, bytes) else name
                zLevelBase = int(getattr(f, p + "zLevelBase"))
                extType = getattr(f, p + "extType")
                extType = (extType.decode() if isinstance(extType, bytes) else extType).strip()
                if extType == "absXsec":  # :216-236 a gas: absorption only
                    if commonD.numConc is None:
                        raise McbratError("read_SSPTable: Error reading scalar fields from file")
                    gasComp += 1
                    xsec = np.array(f.variables[p + "xsec"][:], np.float64)[lambdaIndex - 1]  # Fortran (z, lambda)
                    ext = xsec * commonD.numConc[0, 0, :] * 1000.0  # m^2 x m^-3 -> km^-1
                    ssa = np.zeros(nz)
                    pfi = np.ones(nz, np.int32)
                    table = PhaseFunctionTable([PhaseFunction(legendreCoefficients=np.zeros(2, np.float32))],
                                               np.zeros(1, np.float32), "Molecular Absorption")
                elif extType == "volExt":  # :237-303
                    if commonD.massConc is None or commonD.Reff is None:
                        raise McbratError("read_SSPTable: Error reading scalar fields from file")
                    key = np.array(f.variables[p + "phaseFunctionKeyT"][:], np.float32)
                    extT = np.array(f.variables[p + "ExtinctionT"][:], np.float64)[lambdaIndex - 1]
                    ssaT = np.array(f.variables[p + "SingleScatteringAlbedoT"][:], np.float64)[lambdaIndex - 1]
                    if setup:
                        table = PhaseFunctionTable([PhaseFunction(legendreCoefficients=np.zeros(2, np.float32))],
                                                   np.zeros(1, np.float32), "dummy table")
                    else:
                        table = _read_PhaseFunctionTableNEW(f, p, lambdaIndex)
                    mass = commonD.massConc[comp - gasComp - 1]
                    reff = commonD.Reff[comp - gasComp - 1]
                    key8 = key.astype(np.float64)
                    inside = (mass > 0.0) & (reff < key8.max()) & (reff >= key8.min())
                    if np.any((mass > 0.0) & ~inside):
                        raise McbratError("read_SSPTable: Effective radius outside of table range")
                    il = np.clip(np.searchsorted(key8, reff, side="right"), 1, len(key8) - 1)  # findIndex: key(il) <= Reff < key(il+1)
                    fr = (reff - key8[il - 1]) / (key8[il] - key8[il - 1])
                    ext = np.where(inside, mass * ((1 - fr) * extT[il - 1] + fr * extT[il]), 0.0)
                    ssa = np.where(inside, (1 - fr) * ssaT[il - 1] + fr * ssaT[il], 0.0)
                    pfi = np.ones((nx, ny, nz), np.int32)
                    if not setup:  # the closer phase function :289-296
                        pfi = np.where(inside, np.where(fr < 0.5, il, il + 1), 1).astype(np.int32)
                else:
                    raise McbratError("read_SSPTable: unrecognizable extType")
                dom.addOpticalComponent(name, ext, ssa, pfi, table, zLevelBase=zLevelBase)
                comp += 1
        finally:
            f.close()
    if dom is None:
        raise McbratError("read_SSPTable: no single-scattering-property file given")
    if calcRayl and not setup:  # :323-340
        if commonD.rho is None or commonD.numConc is None:
            raise McbratError("read_SSPTable: Error calculating rayleigh scattering.")
        ext, ssa, pfi, table = calc_RayleighScattering(dom.lambda_um, commonD.rho[0, 0, :], commonD.numConc[0, 0, :])
        dom.addOpticalComponent("Rayleigh Scattering", ext, ssa, pfi, table, zLevelBase=1)
    dom.getOpticalPropertiesByComponent()
    return dom


def read_SolarSource(fileName, nLambda):
    """read_SolarSource, src/emissionAndBroadBandWeights.f95:598-633 -> (sourceFunction, lambdas)."""
    try:
        f = netcdf_file(fileName, "r", mmap=False)
    except (OSError, TypeError, ValueError):
        raise McbratError("read_SolarSource: Can't open file " + str(fileName))
    try:
        if "Lambdas" not in f.dimensions or "Lambdas" not in f.variables or "SourceFunction" not in f.variables:
            raise McbratError("read_SolarSource: %s doesn't look a solar source function file." % fileName)
        if int(f.dimensions["Lambdas"]) != int(nLambda):
            raise McbratError("read_SolarSource: %s dimension of solar source function does not match numLambdas from namelist." % fileName)
        return np.array(f.variables["SourceFunction"][:], np.float64), np.array(f.variables["Lambdas"][:], np.float64)
    finally:
        f.close()


def read_specResponseFunction(fileName, nLambda):
    """read_specResponseFunction, src/emissionAndBroadBandWeights.f95:636-662."""
    try:
        f = netcdf_file(fileName, "r", mmap=False)
    except (OSError, TypeError, ValueError):
        raise McbratError("read_specResponseFunction: Can't open file " + str(fileName))
    try:
        if "Lambdas" not in f.dimensions or "SRF" not in f.variables:
            raise McbratError("read_specResponseFunction: %s doesn't look a spectral response function file." % fileName)
        if int(f.dimensions["Lambdas"]) != int(nLambda):
            raise McbratError("read_specResponseFunction: %s dimension of spectral response function does not match numLambdas from namelist." % fileName)
        return np.array(f.variables["SRF"][:], np.float64)
    finally:
        f.close()
