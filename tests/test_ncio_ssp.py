"""The file family the current driver reads (Drivers/monteCarloDriver.f95:299, :936): read_Common + read_SSPTable
(src/opticalProperties.f95:347-451, :147-345), and the solar source / instrument response files
(src/emissionAndBroadBandWeights.f95:598-662).  The reference tree holds no sample of these files, so the test
writes them in exactly the layout the readers ask for -- dimension, variable and attribute names spelled as THEY
spell them ("x-edges", "Component1_ExtinctionT", Fortran (z, lambda) = NetCDF (lambda, z)) -- and checks the domain
that comes back against the arithmetic of the reader's source text.  CPU only."""
import numpy as np
import pytest
from scipy.io import netcdf_file

from mcbrat3d_amd import ncio
from mcbrat3d_amd._capi import McbratError

NX, NY, NZ, NLAM = 4, 3, 5, 3
FREQ = ncio.LIGHT_SPD * 1e6 / np.array([0.5, 0.8, 1.6])  # Hz for 0.5, 0.8, 1.6 um


def write_common(path, rng, with_density=True, pressure_profile=True):
    f = netcdf_file(path, "w", version=2)
    for name, n in (("x-edges", NX + 1), ("y-edges", NY + 1), ("z-edges", NZ + 1), ("x-grid", NX), ("y-grid", NY), ("z-grid", NZ),
                    ("nonGasComps", 2)):
        f.createDimension(name, n)
    xe, ye, ze = 0.1 * np.arange(NX + 1), 0.2 * np.arange(NY + 1), np.array([0.0, 0.3, 0.7, 1.2, 2.0, 3.5])
    for name, a in (("x-edges", xe), ("y-edges", ye), ("z-edges", ze)):
        f.createVariable(name, "d", (name,))[:] = a
    temps = 220.0 + 60.0 * rng.random((NX, NY, NZ))
    f.createVariable("Temperatures", "d", ("z-grid", "y-grid", "x-grid"))[:] = temps.transpose(2, 1, 0)
    if pressure_profile:
        prs = np.array([950.0, 800.0, 650.0, 500.0, 300.0])
        f.createVariable("Pressures", "d", ("z-grid",))[:] = prs
        prs3 = np.broadcast_to(prs, (NX, NY, NZ))
    else:
        prs3 = 300.0 + 700.0 * rng.random((NX, NY, NZ))
        f.createVariable("Pressures", "d", ("z-grid", "y-grid", "x-grid"))[:] = prs3.transpose(2, 1, 0)
    mass = rng.random((2, NX, NY, NZ)) * (rng.random((2, NX, NY, NZ)) < 0.6)
    reff = 5.0 + 14.9 * rng.random((2, NX, NY, NZ))
    f.createVariable("massConc", "d", ("z-grid", "y-grid", "x-grid", "nonGasComps"))[:] = mass.transpose(3, 2, 1, 0)
    f.createVariable("Reff", "d", ("z-grid", "y-grid", "x-grid", "nonGasComps"))[:] = reff.transpose(3, 2, 1, 0)
    rho = np.array([1.1, 0.95, 0.8, 0.6, 0.35])
    if with_density:
        f.createVariable("Density", "d", ("z-grid",))[:] = rho
    f.close()
    return dict(xe=xe, ye=ye, ze=ze, temps=temps, prs=prs3, mass=mass, reff=reff, rho=rho)


def write_ssp(path, comps, albedo):
    """comps: list of dicts {name, zLevelBase, extType, + xsec[nz, nlam] | key, extT[nkey, nlam], ssaT, legendre | angles/values}."""
    f = netcdf_file(path, "w", version=2)
    f.createDimension("f_grid_nelem", NLAM)
    f.createVariable("f_grid", "d", ("f_grid_nelem",))[:] = FREQ
    f.createVariable("surfaceAlbedo", "d", ("f_grid_nelem",))[:] = albedo
    f.numberOfComponents = np.int32(len(comps))
    for i, c in enumerate(comps, start=1):
        p = "Component%d_" % i
        f.__setattr__(p + "Name", c["name"])
        f.__setattr__(p + "zLevelBase", np.int32(c["zLevelBase"]))
        f.__setattr__(p + "extType", c["extType"])
        if c["extType"] == "absXsec":
            if "z-grid" not in f.dimensions:
                f.createDimension("z-grid", NZ)
            f.createVariable(p + "xsec", "d", ("f_grid_nelem", "z-grid"))[:] = c["xsec"].T  # Fortran (z, lambda)
            continue
        nk = len(c["key"])
        f.createDimension(p + "phaseFunctionNumber", nk)
        f.createVariable(p + "phaseFunctionKeyT", "f", (p + "phaseFunctionNumber",))[:] = c["key"]
        f.createVariable(p + "ExtinctionT", "d", ("f_grid_nelem", p + "phaseFunctionNumber"))[:] = c["extT"].T
        f.createVariable(p + "SingleScatteringAlbedoT", "d", ("f_grid_nelem", p + "phaseFunctionNumber"))[:] = c["ssaT"].T
        f.__setattr__(p + "description", "made by the test")
        if "legendre" in c:  # legendre[lam][entry] = coefficient array; stored back to back per wavelength
            f.__setattr__(p + "phaseFunctionStorageType", "LegendreCoefficients")
            length = np.array([[len(c["legendre"][l][e]) for e in range(nk)] for l in range(NLAM)], np.int32)
            ntot = int(length.sum(axis=1).max())
            start = np.zeros((NLAM, nk), np.int32)
            coeffs = np.zeros((NLAM, ntot), np.float32)
            for l in range(NLAM):
                pos = 1
                for e in range(nk):
                    start[l, e] = pos
                    coeffs[l, pos - 1:pos - 1 + length[l, e]] = c["legendre"][l][e]
                    pos += length[l, e]
            f.createDimension(p + "coefficents", ntot)
            f.createVariable(p + "start", "i", ("f_grid_nelem", p + "phaseFunctionNumber"))[:] = start
            f.createVariable(p + "length", "i", ("f_grid_nelem", p + "phaseFunctionNumber"))[:] = length
            f.createVariable(p + "legendreCoefficients", "f", ("f_grid_nelem", p + "coefficents"))[:] = coeffs
        else:
            f.__setattr__(p + "phaseFunctionStorageType", "Angle-Value")
            f.createDimension(p + "scatteringAngle", len(c["angles"]))
            f.createVariable(p + "scatteringAngle", "f", (p + "scatteringAngle",))[:] = c["angles"]
            f.createVariable(p + "phaseFunctionValues", "f", (p + "phaseFunctionNumber", p + "scatteringAngle"))[:] = c["values"]
    f.close()


@pytest.fixture()
def files(tmp_path):
    rng = np.random.default_rng(11)
    common = write_common(str(tmp_path / "phys.nc"), rng)
    key = np.array([5.0, 10.0, 15.0, 20.0], np.float32)
    g = lambda k, l: np.float32(0.7 + 0.01 * k + 0.02 * l)  # noqa: E731
    gas = dict(name="water vapour", zLevelBase=1, extType="absXsec", xsec=1e-27 * (1.0 + rng.random((NZ, NLAM))))
    liquid = dict(name="liquid", zLevelBase=1, extType="volExt", key=key, extT=0.5 + rng.random((4, NLAM)),
                  ssaT=0.9 + 0.1 * rng.random((4, NLAM)),
                  legendre=[[np.array([g(k, l) ** n for n in range(1, 6 + k)], np.float32) for k in range(4)] for l in range(NLAM)])
    ang = np.linspace(0.0, np.pi, 91).astype(np.float32)
    ang[-1] = np.float32(np.pi)
    ice = dict(name="ice", zLevelBase=1, extType="volExt", key=key, extT=0.2 + rng.random((4, NLAM)),
               ssaT=0.8 + 0.2 * rng.random((4, NLAM)), angles=ang,
               values=np.stack([(1.0 + (0.2 + 0.1 * k) * np.cos(ang)) for k in range(4)]).astype(np.float32))
    albedo = np.array([0.05, 0.2, 0.35])
    write_ssp(str(tmp_path / "ssp1.nc"), [gas, liquid], albedo)
    write_ssp(str(tmp_path / "ssp2.nc"), [ice], albedo)
    return tmp_path, common, gas, liquid, ice, albedo


def test_read_common(files):
    tmp, c, *_ = files
    cd = ncio.read_Common(str(tmp / "phys.nc"))
    assert np.array_equal(cd.xPosition, c["xe"]) and np.array_equal(cd.zPosition, c["ze"])
    assert cd.temps.shape == (NX, NY, NZ) and np.array_equal(cd.temps, c["temps"])
    want = (c["prs"] * 100.0 * ncio.AVOGADRO) / (ncio.RSTAR * c["temps"])  # opticalProperties.f95:424
    assert np.allclose(cd.numConc, want, rtol=1e-15)
    assert cd.massConc.shape == (2, NX, NY, NZ) and np.array_equal(cd.massConc, c["mass"]) and np.array_equal(cd.Reff, c["reff"])
    assert cd.rho.shape == (NX, NY, NZ) and np.array_equal(cd.rho[2, 1, :], c["rho"])
    with pytest.raises(McbratError, match="Can't open file"):
        ncio.read_Common(str(tmp / "missing.nc"))
    with pytest.raises(McbratError, match="problem reading dimensions"):  # the OTHER file family (write_Domain's spelling)
        import mcbrat3d_amd as M
        d = M.new_Domain([0, 1.0], [0, 1.0], [0, 1.0])
        d.addOpticalComponent("c", np.ones((1, 1, 1)), np.ones((1, 1, 1)), np.ones((1, 1, 1), np.int32),
                              M.new_PhaseFunctionTable([M.new_PhaseFunction(np.array([0.5], np.float32))]))
        ncio.write_Domain(d, str(tmp / "other.dom"))
        ncio.read_Common(str(tmp / "other.dom"))


@pytest.mark.parametrize("lam_index", [1, 3])
def test_read_ssp_table(files, lam_index):
    tmp, c, gas, liquid, ice, albedo = files
    cd = ncio.read_Common(str(tmp / "phys.nc"))
    dom = ncio.read_SSPTable([str(tmp / "ssp1.nc"), str(tmp / "ssp2.nc"), "", ""], lam_index, cd, setup=False, calcRayl=True)
    k = lam_index - 1
    assert dom.lambda_um == pytest.approx([0.5, 0.8, 1.6][k], rel=1e-12) and dom.surfaceAlbedo == albedo[k]
    assert [q["name"] for q in dom.components] == ["water vapour", "liquid", "ice", "Rayleigh Scattering"]
    # gas: cross section x number concentration of column (1, 1) x 1000 (:224), absorbs only
    gasc = dom.components[0]
    assert np.allclose(gasc["ext"], gas["xsec"][:, k] * cd.numConc[0, 0, :] * 1000.0, rtol=1e-15) and np.all(gasc["ssa"] == 0.0)
    # condensed components: the Reff interval, linear interpolation of extinction and albedo, nearer phase function (:268-296)
    for comp, src, which in ((dom.components[1], liquid, 0), (dom.components[2], ice, 1)):
        mass, reff, key = c["mass"][which], c["reff"][which], src["key"].astype(np.float64)
        ext, ssa, pfi = np.zeros((NX, NY, NZ)), np.zeros((NX, NY, NZ)), np.ones((NX, NY, NZ), np.int32)
        for ix in range(NX):
            for iy in range(NY):
                for iz in range(NZ):
                    if mass[ix, iy, iz] > 0.0:
                        il = int(np.max(np.nonzero(key <= reff[ix, iy, iz])[0])) + 1  # findIndex, 1-based
                        f = (reff[ix, iy, iz] - key[il - 1]) / (key[il] - key[il - 1])
                        ext[ix, iy, iz] = mass[ix, iy, iz] * ((1 - f) * src["extT"][il - 1, k] + f * src["extT"][il, k])
                        ssa[ix, iy, iz] = (1 - f) * src["ssaT"][il - 1, k] + f * src["ssaT"][il, k]
                        pfi[ix, iy, iz] = il if f < 0.5 else il + 1
        assert np.allclose(comp["ext"], ext, rtol=1e-14) and np.allclose(comp["ssa"], ssa, rtol=1e-14)
        assert np.array_equal(comp["pfIndex"], pfi)
    # phase function tables of THIS wavelength
    liq = dom.components[1]["table"]
    assert liq.nEntries == 4 and liq.description == "made by the test"
    for e in range(4):
        assert np.array_equal(liq.phaseFunctions[e].legendreCoefficients, liquid["legendre"][k][e])
        assert liq.phaseFunctions[e].extinction == liquid["extT"][e, k]
    icet = dom.components[2]["table"]
    assert np.array_equal(icet.phaseFunctions[2].value, ice["values"][2]) and np.array_equal(icet.phaseFunctions[0].scatteringAngle, ice["angles"])
    # Rayleigh component from density and number concentration of column (1, 1) (:2052-2086)
    ray = dom.components[3]
    lam = dom.lambda_um
    mr1 = 6.4328E-5 + (2.94981E-2 / (146 - lam ** -2)) + (2.554E-4 / (41 - lam ** -2))
    want = 32.0E27 * 1.060816681 * float(np.float32(np.pi)) ** 3 * c["rho"] ** 2 * mr1 ** 2 / (3.0 * cd.numConc[0, 0, :] * 1.275 ** 2 * lam ** 4)
    assert np.allclose(ray["ext"], want, rtol=1e-14) and np.all(ray["ssa"] == 1.0)
    assert np.allclose(ray["table"].phaseFunctions[0].legendreCoefficients, [0.0, 0.1])
    # the expanded domain is ready for the integrator
    info = dom.getInfo_Domain()
    assert info["numberOfComponents"] == 4 and info["totalExt"].shape == (NZ, NY, NX)
    assert np.allclose(info["cumExt"][3][info["totalExt"] > 0], 1.0)


def test_read_ssp_table_setup_pass_and_errors(files):
    tmp, c, gas, liquid, ice, albedo = files
    cd = ncio.read_Common(str(tmp / "phys.nc"))
    dom = ncio.read_SSPTable(str(tmp / "ssp1.nc"), 2, cd, setup=True, calcRayl=True)  # emission-weighting pass: no tables, no Rayleigh
    assert len(dom.components) == 2 and dom.components[1]["table"].description == "dummy table"
    assert np.all(dom.components[1]["pfIndex"] == 1)
    cd.Reff[0, 0, 0, 0], cd.massConc[0, 0, 0, 0] = 25.0, 1.0  # outside the table's key range
    with pytest.raises(McbratError, match="Effective radius outside of table range"):
        ncio.read_SSPTable(str(tmp / "ssp1.nc"), 2, cd)
    with pytest.raises(McbratError, match="Can't open file"):
        ncio.read_SSPTable(str(tmp / "nope.nc"), 1, cd)
    with pytest.raises(McbratError, match="doesn't look an optical properties file"):
        ncio.read_SSPTable(str(tmp / "phys.nc"), 1, cd)


def test_solar_source_and_response_files(tmp_path):
    f = netcdf_file(str(tmp_path / "sun.nc"), "w")
    f.createDimension("Lambdas", 4)
    f.createVariable("Lambdas", "d", ("Lambdas",))[:] = [0.4, 0.6, 0.9, 1.5]
    f.createVariable("SourceFunction", "d", ("Lambdas",))[:] = [1.7, 1.8, 0.9, 0.3]
    f.createVariable("SRF", "d", ("Lambdas",))[:] = [0.0, 0.5, 1.0, 0.2]
    f.close()
    src, lam = ncio.read_SolarSource(str(tmp_path / "sun.nc"), 4)
    assert np.array_equal(lam, [0.4, 0.6, 0.9, 1.5]) and np.array_equal(src, [1.7, 1.8, 0.9, 0.3])
    assert np.array_equal(ncio.read_specResponseFunction(str(tmp_path / "sun.nc"), 4), [0.0, 0.5, 1.0, 0.2])
    with pytest.raises(McbratError, match="does not match numLambdas"):
        ncio.read_SolarSource(str(tmp_path / "sun.nc"), 5)
    from mcbrat3d_amd import broadband
    cdf, total = broadband.solar_weighting(src, lam, 0.5, ncio.read_specResponseFunction(str(tmp_path / "sun.nc"), 4))
    w = np.array([0.2, 0.25, 0.45, 0.6]) * 0.5 * src * np.array([0.0, 0.5, 1.0, 0.2])  # emissionAndBroadBandWeights.f95:170-200
    assert np.allclose(cdf, np.cumsum(w) / w.sum(), rtol=1e-14) and total == pytest.approx(w.sum(), rel=1e-14)


NML = """&radiativeTransfer
  solarMu = 0.7, solarAzimuth = 15.0, numLambda = 3, calcRayl = .true. /
&monteCarlo
  numPhotonsPerBatch = 3000, numBatches = 6, iseed = 21, nPhaseIntervals = 9001 /
&algorithms /
&output
  reportAbsorptionProfile = .true. /
&fileNames
  physDomainFile = '%(phys)s', SSPfilename = '%(ssp1)s', '%(ssp2)s',
  solarSourceFile = '%(sun)s', outputNetcdfFile = '%(out)s', outputFluxFile = '%(flux)s' /
"""


def write_run_files(tmp, files):
    """Namelist + solar source file for the three wavelengths of the fixture's SSP tables."""
    f = netcdf_file(str(tmp / "sun.nc"), "w")
    f.createDimension("Lambdas", NLAM)
    f.createVariable("Lambdas", "d", ("Lambdas",))[:] = [0.5, 0.8, 1.6]
    f.createVariable("SourceFunction", "d", ("Lambdas",))[:] = [1.9, 1.2, 0.3]
    f.close()
    nml = tmp / "run.nml"
    nml.write_text(NML % dict(phys=tmp / "phys.nc", ssp1=tmp / "ssp1.nc", ssp2=tmp / "ssp2.nc", sun=tmp / "sun.nc",
                              out=tmp / "out.nc", flux=tmp / "flux.out"))
    return nml


def test_driver_namelist_selects_the_ssp_family(files):
    """fileNames: physDomainFile + SSPfilename(1:4) + solarSourceFile (monteCarloDriver.f95:117-121) -> one domain per
    wavelength (host side only; the run itself is tests/test_gpu_spectral.py)."""
    from mcbrat3d_amd import driver_cli
    tmp = files[0]
    cfg = driver_cli.read_namelists(str(write_run_files(tmp, files)))
    assert cfg["numlambda"] == 3 and cfg["sspfilename"] == [str(tmp / "ssp1.nc"), str(tmp / "ssp2.nc")]
    doms = driver_cli.load_domains(cfg)
    assert len(doms) == 3 and [round(d.lambda_um, 6) for d in doms] == [0.5, 0.8, 1.6]
    assert all(len(d.components) == 4 for d in doms) and doms[2].surfaceAlbedo == 0.35
    assert not np.array_equal(doms[0].totalExt, doms[2].totalExt)
