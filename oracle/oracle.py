"""ctypes front-end of the CPU ORACLE (test infrastructure, not product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  It wraps oracle/libmcbrat_oracle.so (built by oracle/Makefile) and adds
the two array-only restatements that need no C:

* optical_properties_by_component -- src/opticalProperties.f95:966-1072
* batch_statistics                -- Drivers/monteCarloDriver.f95:1023-1050, 1188-1228
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAXC = 8


class OrcProblem(C.Structure):
    _fields_ = [
        ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("nc", C.c_int32),
        ("xe", C.c_void_p), ("ye", C.c_void_p), ("ze", C.c_void_p),
        ("totalExt", C.c_void_p), ("cumExt", C.c_void_p), ("ssa", C.c_void_p),
        ("pfIndex", C.c_void_p),
        ("albedo", C.c_double),
        ("invTables", C.c_void_p), ("invOffset", C.c_void_p),
        ("invNSteps", C.c_void_p), ("invNEntries", C.c_void_p),
        ("useRussianRoulette", C.c_int32),
        ("lwFlag", C.c_float),
        ("surfNumX", C.c_int32), ("surfNumY", C.c_int32),
        ("surfXPosition", C.c_void_p), ("surfYPosition", C.c_void_p), ("surfReflectance", C.c_void_p),
    ]


class OrcIntensity(C.Structure):
    _fields_ = [
        ("nDirections", C.c_int32), ("directions", C.c_void_p),
        ("fwdTables", C.c_void_p), ("fwdOrigTables", C.c_void_p), ("fwdOffset", C.c_void_p), ("fwdNAngles", C.c_void_p),
        ("useHybrid", C.c_int32), ("numOrdersOrig", C.c_int32),
        ("useRussianRoulette", C.c_int32), ("zetaMin", C.c_float),
        ("limitContributions", C.c_int32), ("maxContribution", C.c_float),
    ]


class OrcSource(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("solarMu", C.c_float), ("solarAzimuthDeg", C.c_float),
        ("voxelWeights", C.c_void_p), ("fracAtmsPower", C.c_double),
    ]


class OrcRng(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("mti", C.c_int32), ("mt", C.c_uint32 * 624),
        ("seed", C.c_uint64), ("firstPhoton", C.c_uint64),
        ("photon", C.c_uint64), ("event", C.c_uint32), ("cachedBlock", C.c_uint32), ("buf", C.c_uint32 * 4),
        ("ndraws", C.c_uint64),
    ]


class OrcCounters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "legs", "crossings", "collisions", "absorbEvents", "topExits", "surfaceHits",
        "rouletteKills", "rouletteSurvivals", "badPhotons", "surfaceAbsorbed")] + [("draws", C.c_uint64)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


FATE_DTYPE = np.dtype([("fate", "<i4"), ("ix", "<i4"), ("iy", "<i4"), ("iz", "<i4"),
                       ("nScatter", "<i4"), ("nDraws", "<i4"), ("weight", "<f4")])


def build(force=False):
    """Compile libmcbrat_oracle.so (and oracle/_ref when the reference is present)."""
    so = os.path.join(_HERE, "libmcbrat_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("mcbrat_oracle.c", "mcbrat_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "libmcbrat_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_random_real.restype = C.c_float
        L.orc_mt_next_u32.restype = C.c_uint32
        L.orc_accumulate_extinction.restype = C.c_float
        L.orc_lookup_phase_value.restype = C.c_float
        L.orc_compute_rt_intensity.restype = C.c_int64
        L.orc_compute_rt_intensity.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 10
        L.orc_normalize_intensity.argtypes = [C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 3
        L.orc_lookup_phase_value.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.orc_hybrid_phase_functions.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.orc_intensity_directions.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_forward_angles.argtypes = [C.c_int, C.c_void_p]
        L.orc_phase_values_tabulated.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_compute_rt.restype = C.c_int64
        L.orc_compute_rt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 6
        L.orc_normalize.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 4
        L.orc_find_index_double.argtypes = [C.c_double, C.c_void_p, C.c_int, C.c_int]
        L.orc_find_index_real.argtypes = [C.c_float, C.c_void_p, C.c_int, C.c_int]
        L.orc_find_index_mixed.argtypes = [C.c_float, C.c_void_p, C.c_int, C.c_int]
        L.orc_find_cdf_index.argtypes = [C.c_float, C.c_void_p, C.c_int]
        L.orc_emission_weighting.argtypes = [C.c_int] * 4 + [C.c_void_p] * 7 + [C.c_double] * 4 + [C.c_void_p] * 3
        L.orc_philox_init.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.orc_frequency_distribution.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_surface_reflectance.restype = C.c_float
        L.orc_surface_reflectance.argtypes = [C.c_void_p, C.c_double, C.c_double]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ----------------------------------------------------------------------------
# RandomNumbersForMC.f95
# ----------------------------------------------------------------------------
def mt_rng(seed):
    """new_RandomNumberSequence(seed) -- scalar (int) or vector (sequence)."""
    r = OrcRng()
    if np.isscalar(seed):
        lib().orc_mt_init_scalar(C.byref(r), C.c_int32(int(np.int32(np.uint32(seed)))))
    else:
        s = np.asarray(seed, dtype=np.int32)
        lib().orc_mt_init_vector(C.byref(r), _p(s), C.c_int(len(s)))
    return r


def philox_rng(seed, first_photon=0):
    r = OrcRng()
    lib().orc_philox_init(C.byref(r), int(seed), int(first_photon))
    return r


def random_reals(r, n):
    L = lib()
    return np.array([L.orc_random_real(C.byref(r)) for _ in range(n)], dtype=np.float32)


def frequency_distribution(rng, cdf, total_photons, first_draw=0):
    """getFrequencyDistr (src/emissionAndBroadBandWeights.f95:552-572): photons per wavelength."""
    cdf = np.ascontiguousarray(cdf, np.float64)
    out = np.zeros(cdf.size, np.int64)
    lib().orc_frequency_distribution(C.byref(rng), int(first_draw), int(cdf.size), _p(cdf), int(total_photons), _p(out))
    return out


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*[int(x) for x in ctr])
    k = (C.c_uint32 * 2)(*[int(x) for x in key])
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


# ----------------------------------------------------------------------------
# numericUtilities.f95 / phase functions
# ----------------------------------------------------------------------------
def lobatto(n):
    mus = np.zeros(n, np.float32)
    w = np.zeros(n, np.float32)
    lib().orc_lobatto(C.c_int(n), _p(mus), _p(w))
    return mus, w


def legendre(maxL, mus):
    mus = np.ascontiguousarray(mus, np.float32)
    out = np.zeros((len(mus), maxL + 1), np.float32)
    lib().orc_legendre(C.c_int(maxL), C.c_int(len(mus)), _p(mus), _p(out))
    return out.T  # [l, mu] like legendreP(0:maxL, :)


def find_index(value, table, first_guess=0, kind="double"):
    t = np.ascontiguousarray(table, np.float32 if kind == "real" else np.float64)
    f = {"double": lib().orc_find_index_double, "real": lib().orc_find_index_real,
         "mixed": lib().orc_find_index_mixed}[kind]
    return int(f(value, _p(t), len(t), int(first_guess)))


def find_cdf_index(value, table):
    t = np.ascontiguousarray(table, np.float64)
    return int(lib().orc_find_cdf_index(float(value), _p(t), len(t)))


def phase_values_legendre(coef, angles):
    coef = np.ascontiguousarray(coef, np.float32)
    angles = np.ascontiguousarray(angles, np.float32)
    out = np.zeros(len(angles), np.float32)
    lib().orc_phase_values_legendre(C.c_int(len(coef)), _p(coef), C.c_int(len(angles)), _p(angles), _p(out))
    return out


def normalize_phase_function(angles, values):
    angles = np.ascontiguousarray(angles, np.float32)
    values = np.ascontiguousarray(values, np.float32)
    out = np.zeros_like(values)
    lib().orc_normalize_phase_function(C.c_int(len(angles)), _p(angles), _p(values), _p(out))
    return out


def inverse_table_legendre(coef, nsteps):
    coef = np.ascontiguousarray(coef, np.float32)
    out = np.zeros(nsteps, np.float32)
    lib().orc_inverse_table_legendre(C.c_int(len(coef)), _p(coef), C.c_int(nsteps), _p(out))
    return out


def inverse_table_tabulated(angles, values, nsteps):
    """values: as stored by new_PhaseFunction, i.e. already normalised."""
    angles = np.ascontiguousarray(angles, np.float32)
    values = np.ascontiguousarray(values, np.float32)
    out = np.zeros(nsteps, np.float32)
    lib().orc_inverse_table_tabulated(C.c_int(len(angles)), _p(angles), _p(values), C.c_int(nsteps), _p(out))
    return out


# ----------------------------------------------------------------------------
# opticalProperties.f95:966-1072 getOpticalPropertiesByComponent (array-only)
# ----------------------------------------------------------------------------
def optical_properties_by_component(nx, ny, nz, components):
    """components: list of dicts {ext, ssa, pfIndex, zLevelBase(1-based)} whose
    arrays are either (nx,ny,nzc) ['C' order: index [ix,iy,iz]] or (nzc,) for
    horizontally uniform components.  Returns Fortran-ordered flat arrays
    (x fastest) totalExt[nvox], cumExt[nc*nvox], ssa[nc*nvox], pfIndex[nc*nvox]."""
    nc = len(components)
    cum = np.zeros((nc, nz, ny, nx), np.float64)
    ssa = np.zeros((nc, nz, ny, nx), np.float64)
    pfi = np.zeros((nc, nz, ny, nx), np.int32)
    for i, comp in enumerate(components):
        e = np.asarray(comp["ext"], np.float64)
        s = np.asarray(comp["ssa"], np.float64)
        p = np.asarray(comp["pfIndex"], np.int32)
        base = int(comp.get("zLevelBase", 1))
        if e.ndim == 1:  # horizontally uniform :1033-1043
            k = len(e)
            cum[i, base - 1:base - 1 + k] = e[:, None, None]
            ssa[i, base - 1:base - 1 + k] = s[:, None, None]
            pfi[i, base - 1:base - 1 + k] = p[:, None, None]
        else:
            k = e.shape[2]
            cum[i, base - 1:base - 1 + k] = e.transpose(2, 1, 0)
            ssa[i, base - 1:base - 1 + k] = s.transpose(2, 1, 0)
            pfi[i, base - 1:base - 1 + k] = p.transpose(2, 1, 0)
    for i in range(1, nc):  # :1055-1057
        cum[i] = cum[i] + cum[i - 1]
    total = cum[nc - 1].copy()  # :1058
    mask = total > np.finfo(np.float64).tiny  # :1059-1061
    for i in range(nc):
        cum[i][mask] = cum[i][mask] / total[mask]
    return (np.ascontiguousarray(total.reshape(-1)), np.ascontiguousarray(cum.reshape(-1)),
            np.ascontiguousarray(ssa.reshape(-1)), np.ascontiguousarray(pfi.reshape(-1)))


class Problem:
    """Owns the numpy arrays behind an OrcProblem."""

    def __init__(self, xe, ye, ze, totalExt, cumExt, ssa, pfIndex, albedo, inv_tables,
                 use_russian_roulette=True, lw_flag=-1.0, surface=None):
        self.xe = np.ascontiguousarray(xe, np.float64)
        self.ye = np.ascontiguousarray(ye, np.float64)
        self.ze = np.ascontiguousarray(ze, np.float64)
        self.nx, self.ny, self.nz = len(self.xe) - 1, len(self.ye) - 1, len(self.ze) - 1
        nvox = self.nx * self.ny * self.nz
        self.totalExt = np.ascontiguousarray(totalExt, np.float64).reshape(-1)
        self.cumExt = np.ascontiguousarray(cumExt, np.float64).reshape(-1)
        self.ssa = np.ascontiguousarray(ssa, np.float64).reshape(-1)
        self.pfIndex = np.ascontiguousarray(pfIndex, np.int32).reshape(-1)
        assert self.totalExt.size == nvox and self.cumExt.size % nvox == 0
        self.nc = self.cumExt.size // nvox
        assert self.ssa.size == self.nc * nvox and self.pfIndex.size == self.nc * nvox
        assert len(inv_tables) == self.nc
        # inv_tables: list (per component) of arrays [nEntries, nSteps]
        self.inv = [np.ascontiguousarray(t, np.float32).reshape(-1, np.asarray(t).shape[-1]) for t in inv_tables]
        self.invNSteps = np.array([t.shape[1] for t in self.inv], np.int32)
        self.invNEntries = np.array([t.shape[0] for t in self.inv], np.int32)
        offs = np.cumsum([0] + [t.size for t in self.inv[:-1]]).astype(np.int64)
        self.invOffset = offs
        self.invTables = np.concatenate([t.reshape(-1) for t in self.inv]).astype(np.float32)
        for c in range(self.nc):
            sl = self.pfIndex[c * nvox:(c + 1) * nvox]
            assert sl.min() >= 1 and sl.max() <= self.invNEntries[c], "phase function index out of table"
        self.albedo = float(albedo)
        self.c = OrcProblem(self.nx, self.ny, self.nz, self.nc, _p(self.xe), _p(self.ye), _p(self.ze),
                            _p(self.totalExt), _p(self.cumExt), _p(self.ssa), _p(self.pfIndex),
                            self.albedo, _p(self.invTables), _p(self.invOffset), _p(self.invNSteps),
                            _p(self.invNEntries), 1 if use_russian_roulette else 0, float(lw_flag),
                            0, 0, None, None, None)
        if surface is not None:
            self.set_surface(*surface)

    def set_surface(self, reflectance, x_position, y_position):
        """specifyParameters(surfaceBDRF = new_SurfaceDescription(reflectance, xPosition, yPosition)):
        reflectance[numX-1, numY-1] (Fortran index order), positions increasing."""
        self.surfX = np.ascontiguousarray(x_position, np.float64)
        self.surfY = np.ascontiguousarray(y_position, np.float64)
        r = np.asarray(reflectance, np.float32)
        assert r.shape == (self.surfX.size - 1, self.surfY.size - 1)
        self.surfR = np.ascontiguousarray(r.T).reshape(-1)  # x fastest
        self.c.surfNumX, self.c.surfNumY = self.surfX.size, self.surfY.size
        self.c.surfXPosition, self.c.surfYPosition, self.c.surfReflectance = _p(self.surfX), _p(self.surfY), _p(self.surfR)

    def surface_reflectance(self, x, y):
        """computeSurfaceReflectance(surfaceBDRF, xPos, yPos, ...), src/surfaceProperties.f95:119-147."""
        return float(lib().orc_surface_reflectance(C.byref(self.c), float(x), float(y)))

    def grid_flags(self):
        a, b = C.c_int(), C.c_int()
        dx, dy, dz = C.c_double(), C.c_double(), C.c_double()
        lib().orc_grid_flags(C.byref(self.c), C.byref(a), C.byref(b), C.byref(dx), C.byref(dy), C.byref(dz))
        return bool(a.value), bool(b.value), dx.value, dy.value, dz.value


def solar_source(mu0, azimuth_deg):
    return OrcSource(0, float(mu0), float(azimuth_deg), None, 0.0)


class EmissionSource:
    def __init__(self, voxel_weights, frac_atms_power):
        self.vw = np.ascontiguousarray(voxel_weights, np.float64).reshape(-1)
        self.c = OrcSource(1, 0.0, 0.0, _p(self.vw), float(frac_atms_power))


def emission_weighting(problem, temps, lambda_um, sfc_temp, d_lambda=1.0):
    """emission_weightingNEW: returns (voxelWeights CDF, fracAtmsPower, totalFlux)."""
    temps = np.ascontiguousarray(temps, np.float64).reshape(-1)
    vw = np.zeros(problem.nx * problem.ny * problem.nz, np.float64)
    frac, flux = C.c_double(), C.c_double()
    rc = lib().orc_emission_weighting(problem.nx, problem.ny, problem.nz, problem.nc, _p(problem.xe),
                                      _p(problem.ye), _p(problem.ze), _p(temps), _p(problem.totalExt),
                                      _p(problem.cumExt), _p(problem.ssa), problem.albedo,
                                      float(lambda_um), float(sfc_temp), float(d_lambda), _p(vw),
                                      C.byref(frac), C.byref(flux))
    if rc != 0:
        raise RuntimeError("emission_weighting: total power is 0")
    return vw, frac.value, flux.value


def accumulate_extinction(problem, direction, pos, idx, target=None):
    d = (C.c_float * 3)(*[float(np.float32(x)) for x in direction])
    p = (C.c_double * 3)(*[float(x) for x in pos])
    i = (C.c_int32 * 3)(*[int(x) for x in idx])
    ncross = C.c_int64(0)
    acc = lib().orc_accumulate_extinction(C.byref(problem.c), d, p, i, C.c_int(0 if target is None else 1),
                                          C.c_float(0.0 if target is None else float(target)), C.byref(ncross))
    return float(acc), list(p), list(i), ncross.value


def compute_rt(problem, source, rng, n_photons, want_fates=False):
    """computeRT: raw tallies (weight sums) + counters [+ per-photon fates]."""
    ncol = problem.nx * problem.ny
    nvox = ncol * problem.nz
    up, dn, ab = (np.zeros(ncol, np.float32) for _ in range(3))
    vol = np.zeros(nvox, np.float32)
    cnt = OrcCounters()
    fates = np.zeros(n_photons, FATE_DTYPE) if want_fates else None
    src = source.c if hasattr(source, "c") else source
    n = lib().orc_compute_rt(C.addressof(problem.c), C.addressof(src), C.addressof(rng), int(n_photons),
                             _p(up), _p(dn), _p(ab), _p(vol), C.addressof(cnt),
                             _p(fates) if want_fates else None)
    res = {"n": int(n), "fluxUp": up, "fluxDown": dn, "fluxAbsorbed": ab, "volumeAbsorption": vol,
           "counters": cnt.as_dict()}
    if want_fates:
        res["fates"] = fates
    return res


def intensity_directions(mus, phis_deg):
    """makeDirectionCosines(mu, phi*Pi/180) per intensity direction -> [nDir, 3]."""
    mus = np.ascontiguousarray(mus, np.float32)
    phis = np.ascontiguousarray(phis_deg, np.float32)
    out = np.zeros((mus.size, 3), np.float32)
    lib().orc_intensity_directions(int(mus.size), _p(mus), _p(phis), _p(out))
    return out


def forward_angles(n):
    a = np.zeros(int(n), np.float32)
    lib().orc_forward_angles(int(n), _p(a))
    return a


def phase_values_tabulated(st_angles, st_values, angles):
    sa = np.ascontiguousarray(st_angles, np.float32)
    sv = np.ascontiguousarray(st_values, np.float32)
    an = np.ascontiguousarray(angles, np.float32)
    out = np.zeros(an.size, np.float32)
    lib().orc_phase_values_tabulated(int(sa.size), _p(sa), _p(sv), int(an.size), _p(an), _p(out))
    return out


def hybrid_phase_functions(angles, values, width_deg):
    """computeHybridPhaseFunctions on values[nEntries, nAngles]."""
    an = np.ascontiguousarray(angles, np.float32)
    v = np.ascontiguousarray(values, np.float32).reshape(-1, an.size)
    out = np.zeros_like(v)
    lib().orc_hybrid_phase_functions(int(an.size), int(v.shape[0]), _p(an), _p(v), float(width_deg), _p(out))
    return out


def lookup_phase_value(table, angle):
    t = np.ascontiguousarray(table, np.float32)
    return float(lib().orc_lookup_phase_value(_p(t), int(t.size), float(angle)))


class Intensity:
    """Owns the arrays behind an OrcIntensity: directions, forward tables, variance-reduction choices
    (specifyParameters :1046-1292 + tabulateForwardPhaseFunctions)."""

    def __init__(self, mus, phis_deg, fwd_tables, fwd_orig_tables=None, use_hybrid=False, num_orders_orig=0,
                 use_russian_roulette=False, zeta_min=0.3, limit_contributions=False, max_contribution=3.4e38):
        self.mus = np.ascontiguousarray(mus, np.float32)
        self.phis = np.ascontiguousarray(phis_deg, np.float32)
        self.dirs = intensity_directions(self.mus, self.phis)
        self.nDir = int(self.mus.size)
        self.fwd = [np.ascontiguousarray(t, np.float32).reshape(-1, np.asarray(t).shape[-1]) for t in fwd_tables]
        orig = fwd_tables if fwd_orig_tables is None else fwd_orig_tables
        self.orig = [np.ascontiguousarray(t, np.float32).reshape(-1, np.asarray(t).shape[-1]) for t in orig]
        self.nAngles = np.array([t.shape[1] for t in self.fwd], np.int32)
        self.offset = np.cumsum([0] + [t.size for t in self.fwd[:-1]]).astype(np.int64)
        self.tab = np.concatenate([t.reshape(-1) for t in self.fwd]).astype(np.float32)
        self.tabOrig = np.concatenate([t.reshape(-1) for t in self.orig]).astype(np.float32)
        self.c = OrcIntensity(self.nDir, _p(self.dirs), _p(self.tab), _p(self.tabOrig), _p(self.offset), _p(self.nAngles),
                              int(bool(use_hybrid)), int(num_orders_orig), int(bool(use_russian_roulette)),
                              float(zeta_min), int(bool(limit_contributions)), float(max_contribution))


def compute_rt_intensity(problem, source, rng, n_photons, inten):
    """computeRT with radiance: raw sums incl. intensity[nDir, ny*nx] (+ byComponent, excess)."""
    ncol = problem.nx * problem.ny
    nvox = ncol * problem.nz
    up, dn, ab = (np.zeros(ncol, np.float32) for _ in range(3))
    vol = np.zeros(nvox, np.float32)
    I = np.zeros((inten.nDir, ncol), np.float32)
    byc = np.zeros((problem.nc + 1, inten.nDir, ncol), np.float32)
    exc = np.zeros((problem.nc + 1, inten.nDir), np.float32)
    cnt = OrcCounters()
    src = source.c if hasattr(source, "c") else source
    n = lib().orc_compute_rt_intensity(C.addressof(problem.c), C.addressof(src), C.addressof(rng), int(n_photons),
                                       _p(up), _p(dn), _p(ab), _p(vol), C.addressof(cnt), None,
                                       C.addressof(inten.c), _p(I), _p(byc), _p(exc))
    return {"n": int(n), "fluxUp": up, "fluxDown": dn, "fluxAbsorbed": ab, "volumeAbsorption": vol,
            "intensity": I, "intensityByComponent": byc, "intensityExcess": exc, "counters": cnt.as_dict()}


def compute_radiative_transfer_intensity(problem, source, rng, n_photons, inten):
    """computeRadiativeTransfer + reportResults with radiance: normalised per-batch results;
    intensity[nDir, ny*nx], meanIntensity[nDir] (:980-992)."""
    raw = compute_rt_intensity(problem, source, rng, n_photons, inten)
    norm = normalize(problem, raw["n"], raw)
    I = raw["intensity"].copy()
    byc = raw["intensityByComponent"].copy()
    lib().orc_normalize_intensity(C.addressof(problem.c), C.addressof(inten.c), int(raw["n"]), _p(I), _p(byc),
                                  _p(raw["intensityExcess"]))
    mu, md, ma, prof = report_means(problem, norm)
    ncol = problem.nx * problem.ny
    mean_i = np.array([np.sum(I[d], dtype=np.float32) / np.float32(ncol) for d in range(inten.nDir)], np.float32)
    norm.update(meanFluxUp=mu, meanFluxDown=md, meanFluxAbsorbed=ma, absorbedProfile=prof, n=raw["n"],
                counters=raw["counters"], intensity=I, intensityByComponent=byc, meanIntensity=mean_i)
    return norm


def normalize(problem, n_done, res):
    """computeRadiativeTransfer:328-364 on a compute_rt() result (copies)."""
    out = {k: res[k].copy() for k in ("fluxUp", "fluxDown", "fluxAbsorbed", "volumeAbsorption")}
    lib().orc_normalize(C.addressof(problem.c), int(n_done), _p(out["fluxUp"]), _p(out["fluxDown"]),
                        _p(out["fluxAbsorbed"]), _p(out["volumeAbsorption"]))
    return out


def report_means(problem, norm):
    mu, md, ma = C.c_float(), C.c_float(), C.c_float()
    prof = np.zeros(problem.nz, np.float32)
    lib().orc_report_means(C.byref(problem.c), _p(norm["fluxUp"]), _p(norm["fluxDown"]),
                           _p(norm["fluxAbsorbed"]), _p(norm["volumeAbsorption"]), C.byref(mu),
                           C.byref(md), C.byref(ma), _p(prof))
    return mu.value, md.value, ma.value, prof


def compute_radiative_transfer(problem, source, rng, n_photons):
    """computeRadiativeTransfer + reportResults: normalised per-batch results."""
    raw = compute_rt(problem, source, rng, n_photons)
    norm = normalize(problem, raw["n"], raw)
    mu, md, ma, prof = report_means(problem, norm)
    norm.update(meanFluxUp=mu, meanFluxDown=md, meanFluxAbsorbed=ma, absorbedProfile=prof,
                n=raw["n"], counters=raw["counters"])
    return norm


# ----------------------------------------------------------------------------
# Drivers/monteCarloDriver.f95:1023-1050 (moments) and :1188-1228 (mean/stderr)
# ----------------------------------------------------------------------------
def batch_statistics(batches, solar_flux=1.0):
    """batches: list of (n_photons, value-array).  Photon-weighted first/second
    moments, then mean = F*S1/N and stderr = sqrt(max(0, F^2*S2/N - mean^2)/(B-1))."""
    s1 = None
    s2 = None
    ntot = 0
    for n, x in batches:
        x = np.asarray(x, np.float64)
        s1 = n * x if s1 is None else s1 + n * x
        s2 = n * x * x if s2 is None else s2 + n * x * x
        ntot += n
    nb = len(batches)
    mean = solar_flux * s1 / ntot
    var = np.maximum(0.0, solar_flux ** 2 * s2 / ntot - mean ** 2)
    err = np.sqrt(var / max(nb - 1, 1))
    return mean, err
