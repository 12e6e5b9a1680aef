"""Synthetic domains shared by the tests, bench.py and smoke(): the I3RC-type
configurations of SURVEY.md section 8d, built as plain numpy arrays (no reference
code, no files).  Everything is returned as a dict of host arrays so that the same
case can be handed to the oracle (oracle/oracle.py) and to the product
(mcbrat3d_amd) without either seeing the other."""
import numpy as np


def hg_legendre(g, n):
    """chi_l = g**l, l = 1..n, rounded to float32 the way the reference's
    generators get them (`g**(/ (i, i = 1, n) /)` with default-real g,
    Domain-Files/i3rcStepCloud.f95:55): double power of the float32 g."""
    g32 = float(np.float32(g))
    return np.array([g32 ** l for l in range(1, n + 1)], dtype=np.float64).astype(np.float32)


def plane_parallel(ssa=1.0, tau=0.5, nz=32, height=0.25, size=0.5, g=0.85, nleg=64):
    """Config 1: Domain-Files/planeParallel.f95:27-36,75-78 (1x1x32, tau 0.5)."""
    xe = np.array([0.0, size])
    ye = np.array([0.0, size])
    ze = (height / nz) * np.arange(nz + 1)
    ext = np.full((1, 1, nz), tau / height)
    return dict(name="planeParallel", xe=xe, ye=ye, ze=ze,
                components=[dict(ext=ext, ssa=np.full_like(ext, ssa), pfIndex=np.ones(ext.shape, np.int32),
                                 legendre=[hg_legendre(g, nleg)])],
                albedo=0.0)


def step_cloud(ssa=0.99, g=0.85, nleg=64):
    """Config 2: I3RC case 1 step cloud, Domain-Files/i3rcStepCloud.f95:27-35,62-78,
    in km as the integrator expects (32x1x32, tau 2 | 18, Lx = Ly = 0.5, H = 0.25)."""
    nx, ny, nz = 32, 1, 32
    xe = 0.015625 * np.arange(nx + 1)
    ye = np.array([0.0, 0.5])
    ze = 0.0078125 * np.arange(nz + 1)
    ext = np.zeros((nx, ny, nz))
    ext[:16] = 2.0 / 0.25
    ext[16:] = 18.0 / 0.25
    return dict(name="i3rcStepCloud", xe=xe, ye=ye, ze=ze,
                components=[dict(ext=ext, ssa=np.full_like(ext, ssa), pfIndex=np.ones(ext.shape, np.int32),
                                 legendre=[hg_legendre(g, nleg)])],
                albedo=0.0)


def _smooth_field(rng, n, slope=-5.0 / 3.0):
    """Gaussian random field with an isotropic k**slope power spectrum, unit variance."""
    k = np.fft.fftfreq(n) * n
    kk = np.sqrt(k[:, None] ** 2 + k[None, :] ** 2)
    kk[0, 0] = 1.0
    amp = kk ** (slope / 2.0 - 0.5)
    amp[0, 0] = 0.0
    f = np.fft.ifft2(amp * np.fft.fft2(rng.standard_normal((n, n)))).real
    return (f - f.mean()) / f.std()


def landsat_like(n=128, nz=64, sigma=0.8, mean_tau=10.0, seed=20240601, ssa_cloud=0.99, n_entries=24,
                 nleg=64, regular=False, rayleigh=True, albedo=0.0):
    """Configs 3/5: synthetic 'Landsat-like' 3-D cloud field (the
    real scene files are not in the reference tree).  128x128x64, dx = dy = 30 m,
    irregular dz 20-40 m above z0 = 0, lognormal optical depth with a k^-5/3
    spectrum, cloud thickness ~ sqrt(tau), multi-entry HG phase table keyed by an
    effective-radius proxy, optional 1-D Rayleigh component (Legendre (0, 0.1),
    opticalProperties.f95:2075-2082)."""
    rng = np.random.default_rng(seed)
    dx = 0.03125 if regular else 0.03
    xe = dx * np.arange(n + 1)
    ye = dx * np.arange(n + 1)
    if regular:
        ze = 0.03125 * np.arange(nz + 1)
    else:
        dz = 0.02 + 0.02 * (np.arange(nz) / (nz - 1.0))
        ze = np.concatenate([[0.0], np.cumsum(dz)])
    tau = np.exp(sigma * _smooth_field(rng, n) + np.log(mean_tau) - 0.5 * sigma ** 2)
    base = 8  # cloud base layer (about 200 m)
    thick = np.clip(np.round(12.0 * np.sqrt(tau / mean_tau)), 2, 48).astype(int)
    ext = np.zeros((n, n, nz))
    pfi = np.ones((n, n, nz), np.int32)
    reff = _smooth_field(rng, n)
    for i in range(n):
        for j in range(n):
            k0, k1 = base, min(base + thick[i, j], nz - 2)
            if k1 <= k0:
                continue
            depth = ze[k1] - ze[k0]
            ext[i, j, k0:k1] = tau[i, j] / depth
            # effective radius grows with height in the cloud: entry index by layer + column offset
            frac = (np.arange(k0, k1) - k0) / max(k1 - k0 - 1, 1)
            e = 1 + np.clip(np.round((n_entries - 1) * (0.15 + 0.6 * frac + 0.08 * reff[i, j])), 0, n_entries - 1)
            pfi[i, j, k0:k1] = e.astype(np.int32)
    gs = np.linspace(0.80, 0.87, n_entries)
    comps = [dict(ext=ext, ssa=np.where(ext > 0, ssa_cloud, 0.0), pfIndex=pfi,
                  legendre=[hg_legendre(g, nleg) for g in gs])]
    if rayleigh:
        zmid = 0.5 * (ze[1:] + ze[:-1])
        ray = 0.012 * np.exp(-zmid / 8.0)  # km^-1, ~0.5 um
        comps.append(dict(ext=ray, ssa=np.ones(nz), pfIndex=np.ones(nz, np.int32),
                          legendre=[np.array([0.0, 0.1], np.float32)]))
    return dict(name="landsatLike%dx%dx%d" % (n, n, nz), xe=xe, ye=ye, ze=ze, components=comps, albedo=albedo)


def homog_lw(n=20, ext=5.0, ssa=0.5, g=0.85, nleg=12, temp=280.0, sfc_temp=300.0, albedo=0.1, lam=10.0):
    """Config 4 (one wavelength of it): Domain-Files/homogBBDomain.f95:39-66,
    isothermal homogeneous 20x20x20, 0.1 km cells, thermal emission."""
    xe = 0.1 * np.arange(n + 1)
    e = np.full((n, n, n), ext)
    return dict(name="homogLW%d" % n, xe=xe, ye=xe.copy(), ze=xe.copy(),
                components=[dict(ext=e, ssa=np.full_like(e, ssa), pfIndex=np.ones(e.shape, np.int32),
                                 legendre=[hg_legendre(g, nleg)])],
                albedo=albedo, temps=np.full((n, n, n), temp), sfc_temp=sfc_temp, lambda_um=lam)


def oracle_problem(case, nsteps=10001, use_russian_roulette=True, lw_flag=-1.0):
    """Hand a case to the ORACLE (tests only)."""
    from oracle import oracle as O
    nx, ny, nz = len(case["xe"]) - 1, len(case["ye"]) - 1, len(case["ze"]) - 1
    tot, cum, ssa, pfi = O.optical_properties_by_component(nx, ny, nz, case["components"])
    tables = []
    for comp in case["components"]:
        if "tabulated" in comp:
            tables.append(np.stack([O.inverse_table_tabulated(a, O.normalize_phase_function(a, v), nsteps)
                                    for a, v in comp["tabulated"]]))
        else:
            tables.append(np.stack([O.inverse_table_legendre(c, nsteps) for c in comp["legendre"]]))
    # case["surface"] = (reflectance[numX-1, numY-1], xPosition, yPosition): specifyParameters(surfaceBDRF=)
    return O.Problem(case["xe"], case["ye"], case["ze"], tot, cum, ssa, pfi, case["albedo"], tables,
                     use_russian_roulette=use_russian_roulette, lw_flag=lw_flag, surface=case.get("surface"))


def oracle_intensity(case, mus, phis_deg, n_angles=9001, hybrid_width=None, **kw):
    """Intensity set-up for the ORACLE (tests only): forward tables of every component at n_angles equally spaced
    angles (tabulateForwardPhaseFunctions), hybrid versions when hybrid_width (degrees) is given."""
    from oracle import oracle as O
    angles = O.forward_angles(n_angles)
    orig = []
    for comp in case["components"]:
        if "tabulated" in comp:
            orig.append(np.stack([O.phase_values_tabulated(a, O.normalize_phase_function(a, v), angles)
                                  for a, v in comp["tabulated"]]))
        else:
            orig.append(np.stack([O.phase_values_legendre(c, angles) for c in comp["legendre"]]))
    if hybrid_width:
        tabs = [O.hybrid_phase_functions(angles, t, hybrid_width) for t in orig]
        return O.Intensity(mus, phis_deg, tabs, orig, use_hybrid=True, **kw)
    return O.Intensity(mus, phis_deg, orig, None, **kw)


def product_domain(case):
    """Hand a case to the PRODUCT through its reference-shaped host interface."""
    import mcbrat3d_amd as M
    dom = M.new_Domain(case["xe"], case["ye"], case["ze"], temps=case.get("temps"),
                       surfaceAlbedo=case["albedo"], lambda_um=case.get("lambda_um", 0.0))
    for i, comp in enumerate(case["components"]):
        if "tabulated" in comp:
            table = M.new_PhaseFunctionTable([M.new_PhaseFunction(a, v) for a, v in comp["tabulated"]])
        else:
            table = M.new_PhaseFunctionTable([M.new_PhaseFunction(c) for c in comp["legendre"]])
        dom.addOpticalComponent("component%d" % (i + 1), comp["ext"], comp["ssa"], comp["pfIndex"], table,
                                zLevelBase=comp.get("zLevelBase", 1))
    dom.getOpticalPropertiesByComponent()
    return dom


def product_surface(case):
    """The case's surface description for the PRODUCT (None: the domain's albedo)."""
    import mcbrat3d_amd as M
    if case.get("surface") is None:
        return None
    refl, x, y = case["surface"]
    return M.new_SurfaceDescription(np.asarray(refl, np.float32)[None], x, y)


def patchy_surface(case, nxs=5, nys=3, seed=8):
    """A copy of the case with a reflecting surface of nxs x nys patches (reflectances 0 .. 0.9, one of them black)
    on positions of its own that span the domain."""
    rng = np.random.default_rng(seed)
    out = dict(case)
    x = np.linspace(case["xe"][0], case["xe"][-1], nxs + 1)
    y = np.linspace(case["ye"][0], case["ye"][-1], nys + 1)
    x[1:-1] += rng.uniform(-0.3, 0.3, nxs - 1) * (x[1] - x[0])
    y[1:-1] += rng.uniform(-0.3, 0.3, nys - 1) * (y[1] - y[0])
    refl = rng.uniform(0.05, 0.9, (nxs, nys)).astype(np.float32)
    refl[0, 0] = 0.0
    out["surface"] = (refl, x, y)
    return out


def radar_like(n=128, nz=64, seed=20240602):
    """Config 5: optically thick, strongly absorbing 3-D field: as landsat_like with a
    heavier tail (sigma 1.2, tau up to ~100+) and omega0 = 0.9, so that Russian roulette is played often."""
    return landsat_like(n=n, nz=nz, sigma=1.2, mean_tau=20.0, seed=seed, ssa_cloud=0.9, rayleigh=True)


def tabulated_two_lobe(n_angles=361):
    """An angle/value phase function (the storage Mie tables use, src/scatteringPhaseFunctions.f95:104-164):
    forward HG lobe plus a weak backward lobe, tabulated on a uniform angle grid."""
    ang = np.linspace(0.0, np.pi, n_angles).astype(np.float32)
    ang[-1] = np.float32(np.pi)
    mu = np.cos(ang.astype(np.float64))
    hg = lambda g: (1 - g * g) / (1 + g * g - 2 * g * mu) ** 1.5  # noqa: E731
    return ang, (0.9 * hg(0.8) + 0.1 * hg(-0.4)).astype(np.float32)


def stretched_grid_cloud(nx=24, ny=10, nz=18, seed=3):
    """Genuinely non-uniform x, y and z spacing (geometric stretching), lognormal extinction, two
    angle/value phase-function entries + one Legendre entry in separate components."""
    rng = np.random.default_rng(seed)
    xe = np.concatenate([[0.0], np.cumsum(0.02 * 1.06 ** np.arange(nx))])
    ye = np.concatenate([[0.0], np.cumsum(0.05 * 0.95 ** np.arange(ny))])
    ze = np.concatenate([[0.0], np.cumsum(0.03 * 1.1 ** np.arange(nz))])
    ext = np.exp(rng.normal(np.log(4.0), 0.9, (nx, ny, nz)))
    ext[:, :, :2] = 0.0  # clear layers at the bottom
    pfi = rng.integers(1, 3, (nx, ny, nz)).astype(np.int32)
    ang, val = tabulated_two_lobe()
    _, val2 = tabulated_two_lobe()
    return dict(name="stretched", xe=xe, ye=ye, ze=ze, albedo=0.25,
                components=[dict(ext=ext, ssa=np.where(ext > 0, 0.97, 0.0), pfIndex=pfi,
                                 tabulated=[(ang, val), (ang, (val2 * (1 + 0.3 * np.cos(ang) ** 2)).astype(np.float32))]),
                            dict(ext=0.05 * np.ones(nz), ssa=np.ones(nz), pfIndex=np.ones(nz, np.int32),
                                 legendre=[np.array([0.0, 0.1], np.float32)])])
