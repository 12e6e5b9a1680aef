"""The reference's on-disk formats (NetCDF-3), written and read back without a netCDF library:
.dom domain files with Legendre and angle-value phase tables, and the driver's result file."""
import numpy as np
from scipy.io import netcdf_file

from tests import cases


def test_domain_file_round_trip(tmp_path):
    import mcbrat3d_amd as M
    from mcbrat3d_amd import ncio
    case = cases.landsat_like(n=12, nz=10, n_entries=4)
    dom = cases.product_domain(case)
    ang = np.linspace(0, np.pi, 91).astype(np.float32); ang[-1] = np.float32(np.pi)
    tab = M.new_PhaseFunctionTable([M.new_PhaseFunction(ang, (1 + 0.5 * np.cos(ang) ** 2).astype(np.float32)),
                                    M.new_PhaseFunction(ang, np.ones_like(ang))], key=[1.0, 2.0])
    dom.addOpticalComponent("haze", 0.2 * np.ones(4), 0.95 * np.ones(4), np.array([1, 2, 2, 1], np.int32), tab, zLevelBase=3)
    path = ncio.write_Domain(dom, str(tmp_path / "scene.dom"))
    f = netcdf_file(path, "r", mmap=False)  # the reference's names (opticalProperties.f95:1112-1180)
    assert set(["x-Edges", "y-Edges", "z-Edges", "x-Grid", "y-Grid", "z-Grid", "Component3_z-Grid"]) <= set(f.dimensions)
    assert f.variables["Component1_Extinction"].shape == (10, 12, 12)
    assert f.variables["Component1_PhaseFunctionIndex"].typecode() == "h"
    assert f.variables["Component2_Extinction"].shape == (10,)
    assert f.Component1_phaseFunctionStorageType == b"LegendreCoefficients"
    assert f.Component3_phaseFunctionStorageType == b"Angle-Value" and f.numberOfComponents == 3
    f.close()
    back = ncio.read_Domain(path)
    a, b = dom.getInfo_Domain(), back.getInfo_Domain()
    for k in ("xPosition", "yPosition", "zPosition", "totalExt", "cumExt", "ssa"):
        assert np.array_equal(a[k], b[k]), k
    mask = np.broadcast_to(a["totalExt"] > 0, a["phaseFuncI"].shape)
    assert np.array_equal(a["phaseFuncI"][mask], b["phaseFuncI"][mask])
    assert a["albedo"] == b["albedo"] and a["componentNames"] == b["componentNames"]
    for ta, tb in zip(dom.tabulateInversePhaseFunctions(9001), back.tabulateInversePhaseFunctions(9001)):
        assert np.array_equal(ta, tb)


def test_result_file_layout(tmp_path):
    from mcbrat3d_amd import driver, ncio
    nx, ny, nz = 4, 3, 5
    M = 3 + 3 * nx * ny + nz + nx * ny * nz
    rng = np.random.default_rng(1)
    buf = np.zeros(8 + 2 * M)
    buf[0], buf[1] = 1000.0, 10.0
    x = rng.uniform(0.1, 1.0, M)
    buf[8:8 + M] = 1000.0 * x
    buf[8 + M:] = 1000.0 * x * x * 1.01
    st = driver.statistics(driver.unpack_moments(buf, nx, ny, nz), solarFlux=2.0)
    xe, ye, ze = np.arange(nx + 1.0), np.arange(ny + 1.0) * 2, np.arange(nz + 1.0) * 0.5
    out = ncio.writeResults_netcdf(str(tmp_path / "out.nc"), "scene.dom", st, xe, ye, ze, solarFlux=2.0, solarMu=0.5,
                                   reportAbsorptionProfile=True, reportVolumeAbsorption=True)
    f = netcdf_file(out, "r", mmap=False)
    assert f.version_byte == 2  # 64-bit offset, monteCarloDriver.f95:1559
    assert f.variables["fluxUp"].shape == (ny, nx) and f.variables["absorbedVolume"].shape == (nz, ny, nx)
    assert np.allclose(f.variables["x"][:], [0.5, 1.5, 2.5, 3.5]) and np.allclose(f.variables["z"][:], 0.25 + 0.5 * np.arange(nz))
    assert np.allclose(f.variables["fluxDown"][:].T, st["fluxDown"], rtol=1e-6)
    assert np.allclose(f.variables["absorbedVolume_StdErr"][:].transpose(2, 1, 0), st["absorbedVolume_StdErr"], rtol=1e-6)
    assert f.Number_of_batches == 10 and f.Algorithm == b"Ray_tracing" and abs(f.Solar_flux - 2.0) < 1e-12
    assert np.allclose(st["fluxUp"], 2.0 * x[3:3 + nx * ny].reshape(ny, nx).T)
    f.close()


def test_namelist_reader(tmp_path):
    from mcbrat3d_amd import driver_cli
    p = tmp_path / "run.nml"
    p.write_text("""! comment
&radiativeTransfer
  solarMu = 0.5, solarAzimuth = 30.0 ! trailing comment
  LW_flag = -1.
/
&monteCarlo
  numPhotonsPerBatch = 100000, numBatches = 20, iseed = 7 /
&algorithms
  useRayTracing = .true., useRussianRoulette = .FALSE. /
&output
  reportVolumeAbsorption = T /
&fileNames
  physDomainFile = "builtin:i3rcStepCloud", outputNetcdfFile = 'out.nc'
/
""")
    c = driver_cli.read_namelists(str(p))
    assert c["solarmu"] == 0.5 and c["solarazimuth"] == 30.0 and c["lw_flag"] == -1.0
    assert c["numphotonsperbatch"] == 100000 and c["numbatches"] == 20 and c["iseed"] == 7
    assert c["useraytracing"] is True and c["userussianroulette"] is False and c["reportvolumeabsorption"] is True
    assert c["physdomainfile"] == "builtin:i3rcStepCloud" and c["outputnetcdffile"] == "out.nc"
    assert c["nphaseintervals"] == 10001 and c["reportabsorptionprofile"] is False  # reference defaults (:58-99)
