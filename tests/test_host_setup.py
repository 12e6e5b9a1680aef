"""Host-side set-up routines of the product (no GPU needed) against the oracle:
inverse phase-function tables, optical-property expansion, emission weighting, and the
C ABI surface (library loads, exports every declared symbol, fails loudly without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import oracle as O
from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def M():
    from mcbrat3d_amd import build
    build.build()  # hipcc cross-compiles without a GPU
    import mcbrat3d_amd
    return mcbrat3d_amd


def test_abi_exports_every_declared_symbol(M):
    from mcbrat3d_amd import _capi
    L = _capi.lib()
    header = open(os.path.join(ROOT, "include", "mcbrat.h")).read()
    declared = set(re.findall(r"\b(mcbrat_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.mcbrat_abi_version() == 3


def test_no_cpu_fallback(M):
    """Without a HIP device the product must refuse, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(M.McbratError):
        M.new_Integrator(cases.product_domain(cases.plane_parallel()))


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mcbrat3d_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("the oracle", "").replace("oracle's", "").lower() or \
                    not re.search(r"(import|include|from)\s+\S*oracle", src), f


@pytest.mark.parametrize("g,nleg,nsteps", [(0.85, 64, 10001), (0.85, 12, 9001), (0.0, 0, 9001), (0.3, 1, 9001),
                                            (0.87, 299, 10001)])
def test_inverse_table_legendre_matches_oracle(M, g, nleg, nsteps):
    coef = cases.hg_legendre(g, nleg) if nleg else np.zeros(0, np.float32)
    got = M.new_PhaseFunction(coef).inverse_table(nsteps)
    want = O.inverse_table_legendre(coef, nsteps)
    assert np.array_equal(got.view(np.int32), want.view(np.int32))
    assert got[0] == np.float32(np.pi) and got[-1] == 0.0 and np.all(np.diff(got) <= 0)


def test_inverse_table_tabulated_matches_oracle(M):
    ang = np.linspace(0.0, np.pi, 361).astype(np.float32)
    ang[-1] = np.float32(np.pi)
    g = 0.7
    val = ((1 - g * g) / (1 + g * g - 2 * g * np.cos(ang.astype(np.float64))) ** 1.5).astype(np.float32)
    got = M.new_PhaseFunction(ang, val).inverse_table(9001)
    want = O.inverse_table_tabulated(ang, O.normalize_phase_function(ang, val), 9001)
    assert np.array_equal(got.view(np.int32), want.view(np.int32))
    # the table inverts the HG CDF: the median scattering angle of HG(g)
    med = np.arccos((1 + g * g - ((1 - g * g) / (1 - g + 2 * g * 0.5)) ** 2) / (2 * g))
    assert abs(got[4500] - med) < 2e-3


def test_optical_properties_expansion_matches_oracle(M):
    case = cases.landsat_like(n=16, nz=12, n_entries=5)
    dom = cases.product_domain(case)
    info = dom.getInfo_Domain()
    tot, cum, ssa, pfi = O.optical_properties_by_component(16, 16, 12, case["components"])
    assert np.array_equal(info["totalExt"].reshape(-1), tot)
    assert np.array_equal(info["cumExt"].reshape(-1), cum)
    assert np.array_equal(info["ssa"].reshape(-1), ssa)
    mask = np.tile(tot > 0, 2)
    assert np.array_equal(info["phaseFuncI"].reshape(-1)[mask], pfi[mask])
    assert np.all(info["cumExt"][-1][info["totalExt"] > 0] == 1.0)


def test_emission_weighting_matches_oracle(M):
    case = cases.homog_lw(n=8)
    case["temps"] = case["temps"] + np.linspace(-20, 10, 8)[None, None, :]
    dom = cases.product_domain(case)
    w = M.new_Weights(8, 8, 8)
    flux = M.emission_weighting(dom, w, case["sfc_temp"], dLambda=1.0)
    P = cases.oracle_problem(case)
    vw, frac, flux_ref = O.emission_weighting(P, case["temps"].transpose(2, 1, 0).reshape(-1), case["lambda_um"],
                                              case["sfc_temp"], 1.0)
    assert np.array_equal(w.voxelWeights, vw) and w.fracAtmsPower == frac and flux == flux_ref
    assert vw[-1] == 1.0 and np.all(np.diff(vw) >= 0) and 0 < frac < 1
    # Planck sanity: surface term alone is pi * 0.9 * B(300 K, 10 um) = 28.06 W m-2 um-1
    h, c, k = (float(np.float32(x)) for x in (6.62606957e-34, 2.99792458e8, 1.3806488e-23))
    planck300 = 2 * h * c * c / ((1e-5) ** 5 * (np.exp(h * c / (k * 1e-5 * 300.0)) - 1)) / 1e6
    assert abs((1 - frac) * flux - np.pi * 0.9 * planck300) < 1e-9 * flux


def test_error_behaviour_mirrors_reference(M):
    with pytest.raises(M.McbratError, match="solarMu out of bounds"):
        M.new_PhotonStream(0.0, 0.0, numberOfPhotons=10)
    with pytest.raises(M.McbratError, match="solarAzimuth out of bounds"):
        M.new_PhotonStream(1.0, 400.0, numberOfPhotons=10)
    with pytest.raises(M.McbratError, match="increasing"):
        M.new_Domain([0, 1, 1], [0, 1], [0, 1])
    dom = M.new_Domain([0, 1], [0, 1], [0, 1, 2])
    tbl = M.new_PhaseFunctionTable([M.new_PhaseFunction(cases.hg_legendre(0.85, 8))])
    with pytest.raises(M.McbratError, match="phaseFunctionIndex"):
        dom.addOpticalComponent("c", np.ones((1, 1, 2)), np.ones((1, 1, 2)), 2 * np.ones((1, 1, 2), np.int32), tbl)
    with pytest.raises(M.McbratError, match="singleScatteringAlbedo"):
        dom.addOpticalComponent("c", np.ones((1, 1, 2)), 1.5 * np.ones((1, 1, 2)), np.ones((1, 1, 2), np.int32), tbl)


def test_forward_tables_and_hybrid_phase_functions_match_oracle(M):
    """tabulateForwardPhaseFunctions (opticalProperties.f95:1872-1935) and computeHybridPhaseFunctions
    (:1937-2009): the product's own host routines against the oracle's restatement."""
    from mcbrat3d_amd.phase import computeHybridPhaseFunctions
    n = 1801
    angles = O.forward_angles(n)
    coef = cases.hg_legendre(0.95, 300)
    got = M.new_PhaseFunction(coef).forward_table(n)
    ref = O.phase_values_legendre(coef, angles)
    assert np.allclose(got, ref, rtol=2e-6, atol=1e-6 * ref.max())  # same sums, libm cos vs cosf association
    a, v = cases.tabulated_two_lobe()
    got_t = M.new_PhaseFunction(a, v).forward_table(n)
    ref_t = O.phase_values_tabulated(a, O.normalize_phase_function(a, v), angles)
    assert np.allclose(got_t, ref_t, rtol=1e-5, atol=1e-6)
    hyb = computeHybridPhaseFunctions(np.stack([ref, ref_t * 0 + ref]), 7.0)
    hyb_ref = O.hybrid_phase_functions(angles, np.stack([ref, ref]), 7.0)
    assert np.allclose(hyb, hyb_ref, rtol=1e-5)
    assert hyb[0, 0] < ref[0] and np.array_equal(hyb[0, 400:], ref[400:])
    iso = M.new_PhaseFunction(np.zeros(0, np.float32)).forward_table(11)
    assert np.all(iso == 0.5)


def test_surface_description_host_checks_and_oracle_uniform_case():
    """new_SurfaceDescription (src/surfaceProperties.f95:58-115): the reference's checks and texts; and in the
    oracle a uniform surface description of reflectance 1/4 gives what the domain albedo 1/4 gives (float * float
    and the reference's double product agree exactly for a power of two), while a black patch map does not."""
    import mcbrat3d_amd as M
    from oracle import oracle as O
    s = M.new_SurfaceDescription([0.25])
    assert s.isReady_surfaceDescription() and s.BRDFParameters.shape == (1, 1, 1) and s.xPosition[1] > 1e38
    with pytest.raises(M.McbratError, match="Wrong number of parameters"):
        M.new_SurfaceDescription([0.1, 0.2])
    with pytest.raises(M.McbratError, match="incorrect length"):
        M.new_SurfaceDescription(np.zeros((1, 2, 2), np.float32), [0.0, 1.0], [0.0, 1.0, 2.0])
    with pytest.raises(M.McbratError, match="unique, increasing"):
        M.new_SurfaceDescription(np.zeros((1, 2, 1), np.float32), [0.0, 1.0, 1.0], [0.0, 1.0])
    with pytest.raises(M.McbratError, match="between 0 and 1"):
        M.new_SurfaceDescription(np.full((1, 1, 1), 1.5, np.float32), [0.0, 1.0], [0.0, 1.0])
    n = 4000
    case = cases.step_cloud(ssa=0.99)
    case["albedo"] = 0.25
    a = O.compute_radiative_transfer(cases.oracle_problem(case), O.solar_source(0.7, 20.0), O.mt_rng(5), n)
    huge = float(np.finfo(np.float32).max)
    case["albedo"] = 0.9  # must be ignored once a surface description is given (:667-673)
    case["surface"] = (np.full((1, 1), 0.25, np.float32), [0.0, huge], [0.0, huge])
    b = O.compute_radiative_transfer(cases.oracle_problem(case), O.solar_source(0.7, 20.0), O.mt_rng(5), n)
    for k in ("meanFluxUp", "meanFluxDown", "meanFluxAbsorbed"):
        assert a[k] == b[k], k
    patchy = cases.patchy_surface(cases.step_cloud(ssa=0.99))
    c = O.compute_radiative_transfer(cases.oracle_problem(patchy), O.solar_source(0.7, 20.0), O.mt_rng(5), n)
    assert c["meanFluxUp"] != a["meanFluxUp"]
    # energy: what goes up, is absorbed in the medium or by the surface patches makes up the incoming unit
    refl, x, y = patchy["surface"]
    assert 0.0 < c["meanFluxUp"] + c["meanFluxAbsorbed"] < 1.0
