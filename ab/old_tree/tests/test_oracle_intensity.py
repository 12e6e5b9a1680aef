"""Radiance by local estimation in the ORACLE (computeIntensityContribution,
Integrators/monteCarloRadiativeTransfer.f95:1623-1832): known answers and internal consistency.
The reference holds no recorded radiance outputs, so these pin the restatement to physics it must obey."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import cases


def _vacuum(albedo):
    c = cases.plane_parallel(ssa=1.0)
    c["components"][0]["ext"] = np.zeros_like(c["components"][0]["ext"])
    c["albedo"] = albedo
    return c


def test_lambertian_surface_under_vacuum_is_exact():
    """No atmosphere: every photon reaches the surface with weight 1, is reflected with weight A, and sends
    A/pi to every upward direction (normalizedPhaseFunc = 1/Pi, :1691) -- whatever mu0 and the view angle."""
    case = _vacuum(0.3)
    P = cases.oracle_problem(case, nsteps=101)
    for rr in (False, True):
        I = cases.oracle_intensity(case, [1.0, 0.5, 0.2], [0.0, 90.0, 200.0], n_angles=181, use_russian_roulette=rr)
        res = O.compute_radiative_transfer_intensity(P, O.solar_source(0.7, 30.0), O.mt_rng(5), 2000, I)
        if not rr:
            assert np.allclose(res["meanIntensity"], 0.3 / np.pi, rtol=1e-4)  # float32 running sums
        else:  # zeta = pi * A/pi... the contribution is 1/pi >= zetaMin/pi: full contribution, tau = 0 <= tauMax
            assert np.allclose(res["meanIntensity"], 0.3 / np.pi, rtol=1e-4)  # float32 running sums
        assert abs(res["meanFluxUp"] - 0.3) < 3e-5


def test_single_scattering_radiance_matches_the_analytic_value():
    """Thin conservative layer, black surface, sun overhead: I(mu) ~ omega0 P(Theta) / (4 mu) *
    [1 - exp(-tau (1 + 1/mu))] / (1 + 1/mu) + O(tau^2) for reflected radiance (first order of scattering)."""
    case = cases.plane_parallel(ssa=1.0, tau=0.02)
    P = cases.oracle_problem(case, nsteps=2001)
    mus = [1.0, 0.6]
    I = cases.oracle_intensity(case, mus, [0.0, 0.0], n_angles=9001)
    n = 400000
    res = O.compute_radiative_transfer_intensity(P, O.solar_source(1.0, 0.0), O.mt_rng(11), n, I)
    coef = case["components"][0]["legendre"][0]
    for d, mu in enumerate(mus):
        theta = np.arccos(-mu)  # sun travels along -z, the view direction along +mu
        pval = O.phase_values_legendre(coef, np.array([theta], np.float32))[0]  # normalised to integrate to 2 over mu
        m = 1.0 + 1.0 / mu
        first = pval / (4.0 * np.pi * mu) * (1.0 - np.exp(-0.02 * m)) / m
        got = float(res["meanIntensity"][d])
        assert abs(got - first) / first < 0.08, (mu, got, first)   # second order ~ tau, noise ~ 2 %


def test_intensity_roulette_is_unbiased_and_philox_agrees_with_mt():
    case = cases.step_cloud(0.99)
    P = cases.oracle_problem(case, nsteps=2001)
    mus, phis = [1.0, 0.5], [0.0, 180.0]
    n = 60000
    out = {}
    for name, rr, rng in (("mt", False, O.mt_rng(3)), ("mt_rr", True, O.mt_rng(4)), ("px_rr", True, O.philox_rng(9))):
        I = cases.oracle_intensity(case, mus, phis, n_angles=1801, use_russian_roulette=rr, zeta_min=0.3)
        batches = []
        for b in range(6):
            batches.append(O.compute_radiative_transfer_intensity(P, O.solar_source(1.0, 0.0), rng, n // 6, I)["meanIntensity"])
            if rng.mode == 1:
                rng.firstPhoton += n // 6
        b = np.array(batches)
        out[name] = (b.mean(0), b.std(0, ddof=1) / np.sqrt(len(batches)))
    for other in ("mt_rr", "px_rr"):
        z = (out[other][0] - out["mt"][0]) / np.sqrt(out[other][1] ** 2 + out["mt"][1] ** 2)
        assert np.all(np.abs(z) < 4.5), (other, z, out)
    assert np.all(out["mt"][0] > 0.01)


def test_hybrid_phase_function_is_normalised_and_smooth_in_front():
    coef = cases.hg_legendre(0.95, 300)  # (a broad phase function such as g = 0.85 has no transition angle and is kept)
    angles = O.forward_angles(1801)
    vals = O.phase_values_legendre(coef, angles)[None, :]
    hyb = O.hybrid_phase_functions(angles, vals, 7.0)[0]
    mu = np.cos(angles.astype(np.float64))
    integ = lambda v: float(np.sum(0.5 * (v[1:] + v[:-1]) * (mu[:-1] - mu[1:])))  # noqa: E731
    assert abs(integ(hyb.astype(np.float64)) - 2.0) < 5e-3
    k = int(np.argmax(hyb != vals[0]))          # unchanged behind the transition angle
    last = int(np.max(np.nonzero(hyb != vals[0])))
    assert k == 0 and 10 < last < 600
    assert hyb[0] < vals[0, 0]                  # the forward peak is flattened
    assert np.array_equal(hyb[last + 1:], vals[0, last + 1:])


def test_lookup_interpolates_linearly_in_angle():
    t = np.linspace(1.0, 3.0, 11).astype(np.float32)
    assert O.lookup_phase_value(t, 0.0) == pytest.approx(1.0)
    assert O.lookup_phase_value(t, np.pi) == pytest.approx(3.0, rel=1e-6)
    assert O.lookup_phase_value(t, 0.25 * np.pi) == pytest.approx(1.5, rel=1e-5)
