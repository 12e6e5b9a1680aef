"""Known answers that come from physics, not from the oracle: the closed forms a plane-parallel medium has.  The reference
holds no tests (its authors validated "by eye against analytic / plane-parallel results"); here are
the closed forms that are exact and two deterministic solvers written for the purpose, applied to the ORACLE on the CPU
and -- marked gpu -- to the product through the C ABI, so that the two are also held to something neither of them wrote.

1. Beer-Lambert: omega0 = 0, sun at mu0: the surface receives exp(-tau / mu0), nothing leaves the top
   (accumulateExtinctionAlongPath, src/opticalProperties.f95:1696-1812, and nothing else).
2. Conservation: omega0 = 1 over a surface of albedo 1: every photon leaves through the top, with weight exactly 1.
3. Isothermal emission: omega0 = 0, black surface at the temperature of the layer: the layer emits 4 pi B kappa dz per
   area and the surface pi B (emission_weightingNEW, src/emissionAndBroadBandWeights.f95:424-550).  The top receives
   pi B -- the surface's part attenuated by the flux transmission 2 E3(tau), the layer's own pi B (1 - 2 E3(tau)); Kirchhoff
   -- and the surface receives the layer's part only (space sends nothing).  As fractions of the emitted power, with
   e = 1 / (1 + 4 tau): atmosphere 4 tau e, up = e, down = e (1 - 2 E3(tau)), net absorbed by the layer -e (1 - 2 E3(tau))
   (LW_flag > 0 tallies emission as negative absorption, Integrators/monteCarloRadiativeTransfer.f95:504-508).  This
   exercises the power split, the launch positions, the isotropic and the Lambertian launch directions and the free
   paths together.
4. Multiple scattering, isotropic: a slab that scatters isotropically, against the numerical solution of its integral equation
   (Schwarzschild-Milne: S(t) = omega [ e^(-t/mu0) / (4 pi mu0) + 1/2 int S(t') E1(|t - t'|) dt' ], fluxes 2 pi int S E2):
   piecewise-constant source on 1500 cells with the kernel integrated exactly over every cell -- converged to 10^-6, and
   conserving energy to 10^-6 at omega = 1 by itself.  Holds the collision loop, the scattering-angle tables, the
   direction algebra and the roulette (computeRT :703-821) to transport theory.
5. Multiple scattering, forward-peaked: Henyey-Greenstein slabs (g = 0.6 and the I3RC cases' g = 0.85 with its 64 Legendre
   terms) against matrix doubling on 129 Gauss streams -- a second deterministic solver, written independently of the
   first and equal to it to 10^-6 where both apply -- for the phase function the reference SAMPLES, which is not quite
   the series it is given (`sampled_moments`; a property of the reference that the drop-in keeps, measured below).
6. Thermal emission WITH scattering and a temperature profile: the integral equation again, with the source
   (1 - omega) B(tau) + omega J and the black surface's 1/2 B_s E2: power split, top and surface fluxes.
7. Radiance: the formal solution I(mu) = 1/mu int S e^(-t/mu) dt with the isotropic slab's source function; and, for a
   forward-scattering slab, the mean over 12 view azimuths against the reflection operator of the doubling solver; thermal
   radiance (emission seen directly + the scattered field) against the thermal integral equation's formal solution.
8. Structure: eight unlike layers on irregular levels, two components per cell (a scatterer and an absorber, so that the
   component pick decides every collision) and a Lambertian surface, against the layered integral equation with the
   surface's return as one more unknown.
9. Geometry: a homogeneous medium on a stretched 3-D grid under a low sun must be the slab again (every face crossing and
   periodic wrap adding up to nothing), in every column alike; columns 10^4 km wide must each be their own slab.
10. Config 4's kind of problem: an isothermal layer with forward scattering over a warmer grey surface, against matrix
   doubling with Kirchhoff's emission B (1 - r 1 - t 1) and the surface's emission and reflection as one unknown.
"""
import numpy as np
import pytest
from scipy.special import expn

from tests import cases

SEED = 20241005


def _sigma(p, n):
    return float(np.sqrt(max(p * (1.0 - p), 1e-12) / n))


def slab(tau, ssa, albedo=0.0, nz=8):
    case = cases.plane_parallel(ssa=ssa, tau=tau, nz=nz, g=0.0, nleg=2)
    case["albedo"] = albedo
    return case


def isothermal(tau, n=6, temp=285.0):
    depth = 0.1 * n
    return cases.homog_lw(n=n, ext=tau / depth, ssa=0.0, g=0.0, nleg=2, temp=temp, sfc_temp=temp, albedo=0.0)


def isotropic_slab(b, omega, mu0, cells=1500):
    """(reflected, diffusely transmitted, directly transmitted) share of the incident flux for a slab of optical depth b
    that scatters isotropically with single-scattering albedo omega, sun at mu0, black surface."""
    h = b / cells
    edges = np.arange(cells + 1) * h
    tc = edges[:-1] + 0.5 * h
    kern = 0.5 * np.abs(expn(2, np.abs(tc[:, None] - edges[None, :-1])) - expn(2, np.abs(tc[:, None] - edges[None, 1:])))
    kern[np.arange(cells), np.arange(cells)] = 1.0 - expn(2, 0.5 * h)  # the cell around the point itself: 2 * 1/2 * (E2(0) - E2(h/2))
    direct = (np.exp(-edges[:-1] / mu0) - np.exp(-edges[1:] / mu0)) / (4.0 * np.pi * h)  # cell mean of e^(-t/mu0) / (4 pi mu0)
    src = np.linalg.solve(np.eye(cells) - omega * kern, omega * direct)
    up = 2.0 * np.pi * float(np.sum(src * (expn(3, edges[:-1]) - expn(3, edges[1:]))))
    down = 2.0 * np.pi * float(np.sum(src * (expn(3, b - edges[1:]) - expn(3, b - edges[:-1]))))
    return up, down, float(np.exp(-b / mu0))


def doubling_matrices(b, omega, chi, streams=33, halvings=24):
    """(mu, c, r, t): Gauss nodes and weights on (0, 1) and the reflection / transmission operators of a homogeneous slab
    (optical depth b, albedo omega, phase function sum_l (2 l + 1) chi_l P_l with chi_0 = 1) on azimuthally averaged
    intensities at the nodes.  Matrix doubling from a layer of b / 2^halvings: r = dt M^-1 (omega/2) P+- C,
    t = E - dt M^-1 (E - (omega/2) P++ C); two equal layers: R = r + t (E - r r)^-1 r t, T = t (E - r r)^-1 t.
    (`streams` Gauss nodes carry 2 streams - 1 Legendre terms.)"""
    from numpy.polynomial.legendre import leggauss, legval
    x, w = leggauss(streams)
    mu, c = 0.5 * (x + 1.0), 0.5 * w
    chi = np.asarray(chi, np.float64)
    nl = len(chi)
    assert nl <= 2 * streams - 1, "more Legendre terms than the quadrature integrates"
    pl = np.stack([legval(mu, np.eye(nl)[l]) for l in range(nl)])
    coef = (2 * np.arange(nl) + 1) * chi
    ppp = np.einsum("l,li,lj->ij", coef, pl, pl)
    ppm = np.einsum("l,li,lj->ij", coef * (-1.0) ** np.arange(nl), pl, pl)
    minv, cm, e = np.diag(1.0 / mu), np.diag(c), np.eye(streams)
    dt = b / 2.0 ** halvings
    r = dt * minv @ (0.5 * omega * ppm) @ cm
    t = e - dt * minv @ (e - 0.5 * omega * ppp @ cm)
    for _ in range(halvings):
        g = np.linalg.inv(e - r @ r)
        r, t = r + t @ g @ r @ t, t @ g @ t
    return mu, c, r, t


def doubling_slab(b, omega, chi, node, streams=33, halvings=24):
    """(mu0, reflected, transmitted incl. direct) share of the incident flux for the sun along Gauss node `node` (an odd
    count of streams holds mu0 = 0.5 exactly), black surface.  Independent of `isotropic_slab`, and equal to it to 10^-6 on
    isotropic scattering (tested below); 33 streams carry 65 Legendre terms to 7 digits."""
    mu, c, r, t = doubling_matrices(b, omega, chi, streams, halvings)
    inc = np.zeros(streams)
    inc[node] = 1.0 / (2.0 * np.pi * mu[node] * c[node])
    return float(mu[node]), float(2.0 * np.pi * np.sum(mu * c * (r @ inc))), float(2.0 * np.pi * np.sum(mu * c * (t @ inc)))


def thermal_doubling(b, omega, chi, planck_layer, planck_sfc, albedo, streams=33):
    """An isothermal, homogeneous, emitting and anisotropically scattering layer over a Lambertian surface of reflectance
    `albedo` and emissivity 1 - albedo (emission_weightingNEW :478).  Kirchhoff gives the layer's own emission from its
    reflection and transmission: in an enclosure at its temperature every outgoing intensity is B, so it emits
    B (1 - r 1 - t 1) into each direction; the surface sends (1 - albedo) B_s + albedo F / pi upward, F the flux onto it.
    Returns (the layer's share of the emitted power, flux through the top, flux onto the surface) as shares of the emitted
    power 4 pi (1 - omega) B b + pi (1 - albedo) B_s."""
    mu, c, r, t = doubling_matrices(b, omega, chi, streams)
    one = np.ones(streams)
    flux = lambda v: float(2.0 * np.pi * np.sum(mu * c * v))  # noqa: E731
    emit = planck_layer * (one - r @ one - t @ one)
    # F = flux(emit) + I_s flux(r 1),  I_s = (1 - albedo) B_s + albedo F / pi
    f_r, f_t = flux(r @ one), flux(t @ one)
    onto = (flux(emit) + (1.0 - albedo) * planck_sfc * f_r) / (1.0 - albedo * f_r / np.pi)
    i_s = (1.0 - albedo) * planck_sfc + albedo * onto / np.pi
    up = flux(emit) + i_s * f_t
    atm = 4.0 * np.pi * (1.0 - omega) * planck_layer * b
    total = atm + np.pi * (1.0 - albedo) * planck_sfc
    return atm / total, up / total, onto / total


def sampled_moments(chi, terms=256, table=None, q=24, nodes=None):
    """Legendre moments (chi_0 .. chi_{terms-1}) of the phase function the reference actually samples when it is handed a
    Legendre series.  Two of its rules matter, and the drop-in keeps both (its tables are the rules' bit for bit):

    * computeInversePhaseFunction (src/inversePhaseFunctions.f95:100-129) evaluates the series at the n = (number of
      moments) Lobatto nodes only and integrates the values by the trapezoidal rule in mu: it inverts the CDF of the
      series' values at those nodes joined by straight lines -- a polygon.  For the I3RC cases' phase function (g = 0.85,
      64 terms, 64 nodes across a forward peak of 80) that is not the series: asymmetry 0.8518 instead of 0.85, 1.3 %
      less backscatter.  `table=None`: the polygon's moments.
    * computeScatteringAngle (Integrators/monteCarloRadiativeTransfer.f95:1594-1621) finds the table interval with
      int(u * nSteps) + 1 and interpolates with the REMAINDER u - (i - 1) / nSteps taken as the fraction of the interval
      (it is 1 / nSteps of one at most): the angle is, to a part in nSteps, the table entry at the interval's start -- the
      LARGER angle of the two.  A staircase that moves the mean cosine by -1 / nSteps and adds 0.27 % to the backscatter of
      the same phase function at nSteps = 9001.  `table=nSteps`: the moments of that staircase over the polygon's table.

    `nodes=(mu ascending, values)`: an angle / value ("Mie table") phase function, whose native angles the reference takes
    as the polygon's nodes."""
    from numpy.polynomial.legendre import leggauss, legval, Legendre
    if nodes is not None:  # an angle / value phase function: its own angles are the polygon's nodes (:80-88)
        mus, vals = np.asarray(nodes[0], np.float64), np.asarray(nodes[1], np.float64)
    else:
        chi = np.asarray(chi, np.float64)
        n = max(len(chi) - 1, 2)
        mus = np.concatenate([[-1.0], np.sort(Legendre.basis(n - 1).deriv().roots().real), [1.0]])  # computeLobattoTerms
        vals = legval(mus, (2 * np.arange(len(chi)) + 1) * chi)
    if table is None:
        x, w = leggauss(q)
        half = 0.5 * (mus[1:] - mus[:-1])[:, None]
        m = (half * x[None, :] + 0.5 * (mus[1:] + mus[:-1])[:, None])
        f = ((vals[:-1, None] + (vals[1:] - vals[:-1])[:, None] * (m - mus[:-1, None]) / (2.0 * half)) * half * w[None, :]).ravel()
        m = m.ravel()
    else:
        # the polygon's CDF, inverted at the table's probabilities k / (nSteps - 1): within a segment the CDF is a parabola
        cdf = np.concatenate([[0.0], np.cumsum(0.5 * (mus[1:] - mus[:-1]) * (vals[1:] + vals[:-1]))])
        cdf /= cdf[-1]
        vn = vals / (0.5 * np.sum((mus[1:] - mus[:-1]) * (vals[1:] + vals[:-1])))  # density of the normalised CDF
        prob = np.arange(table) / (table - 1.0)
        seg = np.clip(np.searchsorted(cdf, prob, side="right") - 1, 0, len(mus) - 2)
        a = 0.5 * (vn[seg + 1] - vn[seg]) / (mus[seg + 1] - mus[seg])   # c(mu) = cdf_i + v_i d + a d^2, d = mu - mu_i
        b, c = vn[seg], cdf[seg] - prob
        disc = np.sqrt(np.maximum(b * b - 4.0 * a * c, 0.0))
        d = np.where(np.abs(a) > 1e-14, (2.0 * -c) / (b + disc), -c / b)  # (the root that goes to -c / b as a -> 0)
        ang = np.arccos(np.clip(mus[seg] + d, -1.0, 1.0))                  # entries 0 .. nSteps-1: pi .. 0
        ang[-1] = 0.0
        ang[:-1] += (ang[1:] - ang[:-1]) / (2.0 * table)                   # (the mean of the sliver that IS interpolated)
        m = np.cos(ang)
        f = np.full(table, 2.0 / table)                                    # (normalised like the density: (1/2) int p dmu = 1)
    p0, p1 = np.ones_like(m), m.copy()
    mom = np.zeros(terms)
    mom[0], mom[1] = 0.5 * f.sum(), 0.5 * (p1 * f).sum()
    for l in range(2, terms):
        p0, p1 = p1, ((2 * l - 1) * m * p1 - (l - 1) * p0) / l
        mom[l] = 0.5 * (p1 * f).sum()
    return mom / mom[0]


def hg_slab(tau, ssa, g, nleg, nz=16):
    case = cases.plane_parallel(ssa=ssa, tau=tau, nz=nz, g=g, nleg=nleg)
    chi = np.concatenate([[1.0], np.asarray(case["components"][0]["legendre"][0], np.float64)])
    return case, chi


# (optical depth, omega0, g, Legendre terms, Gauss node of the sun out of 129: 64 is mu0 = 0.5): the last three have the
# phase function of the I3RC cases themselves (g = 0.85, 64 terms: Domain-Files/i3rcStepCloud.f95:55)
HG_STREAMS = 129
HG_SLABS = [(2.0, 0.95, 0.6, 48, 64), (1.0, 1.0, 0.85, 64, 124), (4.0, 0.9, 0.85, 64, 20), (18.0, 0.99, 0.85, 64, 64)]

def planck(lambda_um, temp):
    """Planck's function, arbitrary units (only ratios enter)."""
    return 1.0 / (lambda_um ** 5 * np.expm1(14387.77 / (lambda_um * np.asarray(temp, np.float64))))


def thermal_slab(b, omega, planck_layers, planck_sfc, cells_per_layer=150):
    """An emitting, isotropically scattering slab over a black surface: `planck_layers` from the TOP layer down (equal
    optical depth each), tau from the top.  S = (1 - omega) B + omega J, J = 1/2 int S E1 + 1/2 B_s E2(b - tau).
    Returns (the layer's share of the emitted power, flux through the top, flux onto the surface) -- the fluxes as shares
    of the emitted power 4 pi (1 - omega) int B dtau + pi B_s, which is how the integrator reports them."""
    planck_layers = np.asarray(planck_layers, np.float64)
    cells = cells_per_layer * len(planck_layers)
    h = b / cells
    edges = np.arange(cells + 1) * h
    tc = edges[:-1] + 0.5 * h
    kern = 0.5 * np.abs(expn(2, np.abs(tc[:, None] - edges[None, :-1])) - expn(2, np.abs(tc[:, None] - edges[None, 1:])))
    kern[np.arange(cells), np.arange(cells)] = 1.0 - expn(2, 0.5 * h)
    bb = np.repeat(planck_layers, cells_per_layer)
    from_surface = 0.5 * planck_sfc * (expn(3, b - edges[1:]) - expn(3, b - edges[:-1])) / h   # cell mean of 1/2 B_s E2(b - t)
    src = np.linalg.solve(np.eye(cells) - omega * kern, (1.0 - omega) * bb + omega * from_surface)
    atm = 4.0 * np.pi * (1.0 - omega) * float(np.sum(bb)) * h
    total = atm + np.pi * planck_sfc
    up = (2.0 * np.pi * float(np.sum(src * (expn(3, edges[:-1]) - expn(3, edges[1:])))) + 2.0 * np.pi * planck_sfc * float(expn(3, b))) / total
    down = 2.0 * np.pi * float(np.sum(src * (expn(3, b - edges[1:]) - expn(3, b - edges[:-1])))) / total
    return atm / total, up, down


def thermal_radiance(b, omega, planck_layers, planck_sfc, mus, cells_per_layer=150):
    """Radiance leaving the top of `thermal_slab`'s layer along the cosines `mus`, as a share of the emitted power:
    I(mu) = 1/mu int S e^(-t/mu) dt + B_s e^(-b/mu)."""
    planck_layers = np.asarray(planck_layers, np.float64)
    cells = cells_per_layer * len(planck_layers)
    h = b / cells
    edges = np.arange(cells + 1) * h
    tc = edges[:-1] + 0.5 * h
    kern = 0.5 * np.abs(expn(2, np.abs(tc[:, None] - edges[None, :-1])) - expn(2, np.abs(tc[:, None] - edges[None, 1:])))
    kern[np.arange(cells), np.arange(cells)] = 1.0 - expn(2, 0.5 * h)
    bb = np.repeat(planck_layers, cells_per_layer)
    from_surface = 0.5 * planck_sfc * (expn(3, b - edges[1:]) - expn(3, b - edges[:-1])) / h
    src = np.linalg.solve(np.eye(cells) - omega * kern, (1.0 - omega) * bb + omega * from_surface)
    total = 4.0 * np.pi * (1.0 - omega) * float(np.sum(bb)) * h + np.pi * planck_sfc
    return np.array([(float(np.sum(src * (np.exp(-edges[:-1] / m) - np.exp(-edges[1:] / m)))) + planck_sfc * np.exp(-b / m)) / total for m in mus])


def thermal_case(tau, ssa, temps_bottom_up, sfc_temp, lam=10.0):
    n = len(temps_bottom_up)
    case = cases.homog_lw(n=n, ext=tau / (0.1 * n), ssa=ssa, g=0.0, nleg=2, temp=280.0, sfc_temp=sfc_temp, albedo=0.0, lam=lam)
    case["temps"] = np.broadcast_to(np.asarray(temps_bottom_up, np.float64)[None, None, :], case["temps"].shape).copy()
    frac, up, down = thermal_slab(tau, ssa, planck(lam, np.asarray(temps_bottom_up)[::-1]), float(planck(lam, sfc_temp)))
    return case, frac, up, down


# (optical depth, omega0, layer temperatures from the surface up, surface temperature)
THERMAL_SLABS = [(1.0, 0.5, [285.0] * 6, 285.0), (2.0, 0.8, [300.0, 294.0, 287.0, 279.0, 270.0, 260.0], 305.0),
                 (0.5, 0.0, [250.0, 260.0, 270.0, 280.0, 290.0, 300.0], 240.0)]

def isotropic_radiance(b, omega, mu0, mus, cells=1500):
    """Radiance leaving the top of the isotropically scattering slab of `isotropic_slab` along the cosines `mus`, per unit
    incident flux: I(mu) = 1/mu int S(t) e^(-t/mu) dt with the same source function (no azimuth: the scattering forgets it)."""
    h = b / cells
    edges = np.arange(cells + 1) * h
    tc = edges[:-1] + 0.5 * h
    kern = 0.5 * np.abs(expn(2, np.abs(tc[:, None] - edges[None, :-1])) - expn(2, np.abs(tc[:, None] - edges[None, 1:])))
    kern[np.arange(cells), np.arange(cells)] = 1.0 - expn(2, 0.5 * h)
    direct = (np.exp(-edges[:-1] / mu0) - np.exp(-edges[1:] / mu0)) / (4.0 * np.pi * h)
    src = np.linalg.solve(np.eye(cells) - omega * kern, omega * direct)
    return np.array([float(np.sum(src * (np.exp(-edges[:-1] / m) - np.exp(-edges[1:] / m)))) for m in mus])


def layered_isotropic_slab(dtaus, omegas, mu0, albedo=0.0, cells_per_layer=150):
    """The isotropically scattering slab again, now in layers (`dtaus`, `omegas` from the TOP layer down) over a Lambertian
    surface of reflectance `albedo`: (flux through the top, flux onto the surface -- every arrival, as the integrator
    counts them).  The surface is one more unknown: the flux F onto it returns albedo * F isotropically, i.e. the mean
    intensity albedo * F / pi * 1/2 E2(b - tau) in the layer."""
    dtaus, omegas = np.asarray(dtaus, np.float64), np.asarray(omegas, np.float64)
    h = np.repeat(dtaus / cells_per_layer, cells_per_layer)
    om = np.repeat(omegas, cells_per_layer)
    edges = np.concatenate([[0.0], np.cumsum(h)])
    b = edges[-1]
    tc = 0.5 * (edges[:-1] + edges[1:])
    cells = len(h)
    kern = 0.5 * np.abs(expn(2, np.abs(tc[:, None] - edges[None, :-1])) - expn(2, np.abs(tc[:, None] - edges[None, 1:])))
    kern[np.arange(cells), np.arange(cells)] = 1.0 - expn(2, 0.5 * h)
    direct = (np.exp(-edges[:-1] / mu0) - np.exp(-edges[1:] / mu0)) / (4.0 * np.pi * h)
    lamb = 0.5 * (expn(3, b - edges[1:]) - expn(3, b - edges[:-1])) / h / np.pi      # mean intensity per unit flux leaving the surface
    to_top = 2.0 * np.pi * (expn(3, edges[:-1]) - expn(3, edges[1:]))                 # flux through the top per unit source in a cell
    to_sfc = 2.0 * np.pi * (expn(3, b - edges[1:]) - expn(3, b - edges[:-1]))
    # unknowns: the source in every cell and the flux F onto the surface
    a = np.zeros((cells + 1, cells + 1))
    a[:cells, :cells] = np.eye(cells) - om[:, None] * kern
    a[:cells, cells] = -om * albedo * lamb
    a[cells, :cells] = -to_sfc
    a[cells, cells] = 1.0
    rhs = np.concatenate([om * direct, [np.exp(-b / mu0)]])
    sol = np.linalg.solve(a, rhs)
    src, onto = sol[:cells], float(sol[cells])
    up = float(np.sum(src * to_top)) + albedo * onto * 2.0 * float(expn(3, b))
    return up, onto


def doubling_radiance(b, omega, chi, node, view_nodes, streams=129):
    """Azimuthal mean of the radiance leaving the top along the Gauss nodes `view_nodes`, per unit incident flux, sun along
    node `node`: row `view` of the reflection operator applied to the beam."""
    mu, c, r, _ = doubling_matrices(b, omega, chi, streams)
    inc = np.zeros(streams)
    inc[node] = 1.0 / (2.0 * np.pi * mu[node] * c[node])
    out = r @ inc
    return float(mu[node]), [float(mu[v]) for v in view_nodes], np.array([float(out[v]) for v in view_nodes])


RADIANCE_AZIMUTHS = 12   # (the mean over 12 equally spaced azimuths is the azimuthal mean up to the 12th Fourier mode: g^12 = 0.2 % at g = 0.6)

RADIANCE_SLABS = [(1.0, 1.0, 0.6), (3.0, 0.9, 1.0)]
RADIANCE_MUS, RADIANCE_PHIS = [1.0, 0.7, 0.35], [0.0, 120.0, 250.0]

SCATTERING_SLABS = [(1.0, 1.0, 1.0), (2.0, 0.9, 0.5), (0.5, 1.0, 0.3), (4.0, 0.6, 0.8)]


def test_the_integral_equation_solver_conserves_energy_and_has_converged():
    for b, mu0 in ((1.0, 1.0), (0.5, 0.3), (3.0, 0.7)):
        up, down, direct = isotropic_slab(b, 1.0, mu0)
        assert abs(up + down + direct - 1.0) < 3e-6
        assert np.allclose(isotropic_slab(b, 1.0, mu0, cells=600)[:2], (up, down), atol=5e-6)


def test_the_thermal_solver_has_the_closed_form_as_its_limit():
    for tau in (0.25, 1.0, 3.0):
        frac, up, down = thermal_slab(tau, 0.0, [1.0] * 4, 1.0)
        e = 1.0 / (1.0 + 4.0 * tau)
        assert abs(frac - 4.0 * tau * e) < 1e-9 and abs(up - e) < 2e-6 and abs(down - e * (1.0 - 2.0 * float(expn(3, tau)))) < 2e-6


def test_the_layered_solver_reduces_to_the_homogeneous_one_and_conserves_energy():
    up, down, direct = isotropic_slab(2.0, 0.9, 0.5)
    u2, onto = layered_isotropic_slab([0.5, 1.0, 0.5], [0.9, 0.9, 0.9], 0.5, cells_per_layer=400)
    assert abs(u2 - up) < 3e-6 and abs(onto - down - direct) < 3e-6
    u3, onto3 = layered_isotropic_slab([0.3, 1.2, 0.5], [1.0, 1.0, 1.0], 0.7, albedo=0.6)
    assert abs(u3 + (1.0 - 0.6) * onto3 - 1.0) < 3e-5 and onto3 > onto   # (450 cells: the midpoint rule is good to 10^-5 here)


LAYERED = dict(dtaus=[0.05, 0.4, 1.5, 0.2, 0.9, 0.02, 0.6, 0.3], omegas=[1.0, 0.95, 0.999, 0.3, 0.8, 1.0, 0.9, 0.6])


def layered_case(albedo):
    """Eight layers of unlike thickness, extinction and single-scattering albedo (top down as in LAYERED) in TWO components:
    a conservative scatterer and an absorber, mixed per layer -- so that the component pick (computeRT :759-760) decides
    every collision -- both isotropic."""
    dt, om = np.asarray(LAYERED["dtaus"]), np.asarray(LAYERED["omegas"])
    nz = len(dt)
    ze = np.concatenate([[0.0], np.cumsum(np.array([0.03, 0.05, 0.02, 0.08, 0.04, 0.06, 0.02, 0.05]))])
    ext = (dt[::-1] / np.diff(ze))[None, None, :]      # per layer, bottom up
    omb = om[::-1][None, None, :]
    iso = [np.zeros(2, np.float32)]
    comps = [dict(ext=ext * omb, ssa=np.ones_like(ext), pfIndex=np.ones(ext.shape, np.int32), legendre=iso),
             dict(ext=ext * (1.0 - omb), ssa=np.zeros_like(ext), pfIndex=np.ones(ext.shape, np.int32), legendre=iso)]
    return dict(name="layered", xe=np.array([0.0, 0.4]), ye=np.array([0.0, 0.7]), ze=ze, components=comps, albedo=albedo)


@pytest.mark.parametrize("albedo,mu0", [(0.0, 0.8), (0.5, 0.35)])
def test_oracle_layers_two_components_and_a_lambertian_surface(albedo, mu0):
    """Vertical structure (irregular z levels: the bisection launch and the edge-table walk), two components per cell, and
    the Lambertian surface (computeRT :640-700) against the layered integral equation."""
    from oracle import oracle as O
    n = 200000
    up, onto = layered_isotropic_slab(LAYERED["dtaus"], LAYERED["omegas"], mu0, albedo=albedo)
    r = O.compute_radiative_transfer(cases.oracle_problem(layered_case(albedo), nsteps=101), O.solar_source(mu0, 10.0), O.philox_rng(SEED, 0), n)
    assert abs(r["meanFluxUp"] - up) < 6.0 * _sigma(up, n)
    assert abs(r["meanFluxDown"] - onto) < 6.0 * _sigma(min(onto, 0.5), n) * (1.0 + albedo)
    assert abs(r["meanFluxAbsorbed"] - (1.0 - up - (1.0 - albedo) * onto)) < 6.0 * (_sigma(up, n) + _sigma(min(onto, 0.5), n))


def homogeneous_3d(tau, ssa, g, nleg):
    """One medium in 7 x 5 x 12 cells of unlike size (stretched in x, y and z): whatever the grid, the answer is the slab's."""
    xe = np.concatenate([[0.0], np.cumsum(0.02 * 1.35 ** np.arange(7))])
    ye = np.concatenate([[0.0], np.cumsum(0.05 * 0.8 ** np.arange(5))])
    ze = np.concatenate([[0.0], np.cumsum(0.01 * 1.2 ** np.arange(12))])
    ext = np.full((7, 5, 12), tau / ze[-1])
    leg = [cases.hg_legendre(g, nleg)] if g else [np.zeros(2, np.float32)]
    return dict(name="homogeneous3d", xe=xe, ye=ye, ze=ze, albedo=0.0,
                components=[dict(ext=ext, ssa=np.full_like(ext, ssa), pfIndex=np.ones(ext.shape, np.int32), legendre=leg)])


def wide_columns(taus, ssa, width_km=1.0e4):
    """Columns so wide that none knows of its neighbours: each is its own slab (the independent-column limit)."""
    nx, nz = len(taus), 8
    xe = width_km * np.arange(nx + 1)
    ze = 0.03125 * np.arange(nz + 1)
    ext = np.repeat((np.asarray(taus, np.float64) / ze[-1])[:, None, None], nz, axis=2)
    return dict(name="wideColumns", xe=xe, ye=np.array([0.0, width_km]), ze=ze, albedo=0.0,
                components=[dict(ext=ext, ssa=np.full_like(ext, ssa), pfIndex=np.ones(ext.shape, np.int32),
                                 legendre=[np.zeros(2, np.float32)])])


def test_oracle_a_homogeneous_medium_on_a_stretched_3d_grid_is_the_slab():
    """Every x / y / z face crossing, the periodic wraps (a sun at 20 degrees elevation carries a photon through the domain's
    sides many times) and the irregular-grid launch must add up to nothing: accumulateExtinctionAlongPath
    (src/opticalProperties.f95:1696-1812) against the integral equation."""
    from oracle import oracle as O
    n = 200000
    up, down, direct = isotropic_slab(1.5, 0.95, 0.35)
    r = O.compute_radiative_transfer(cases.oracle_problem(homogeneous_3d(1.5, 0.95, 0.0, 0), nsteps=101), O.solar_source(0.35, 57.0),
                                     O.philox_rng(SEED, 0), n)
    assert abs(r["meanFluxUp"] - up) < 6.0 * _sigma(up, n)
    assert abs(r["meanFluxDown"] - down - direct) < 6.0 * _sigma(down + direct, n)
    # ... and in every column alike (the normalisation by column area, reportResults :877-884)
    col = np.asarray(r["fluxUp"], np.float64)
    assert col.size == 35 and np.all(np.abs(col - up) < 6.0 * _sigma(up, n / 35.0) * 3.0)


def test_oracle_wide_columns_are_independent_slabs():
    from oracle import oracle as O
    n = 300000
    taus = [2.0, 18.0, 0.5]
    r = O.compute_radiative_transfer(cases.oracle_problem(wide_columns(taus, 0.99), nsteps=101), O.solar_source(0.7, 0.0), O.philox_rng(SEED, 0), n)
    for i, tau in enumerate(taus):
        up, down, direct = isotropic_slab(tau, 0.99, 0.7)
        assert abs(float(r["fluxUp"][i]) - up) < 6.0 * _sigma(up, n / 3.0)
        assert abs(float(r["fluxDown"][i]) - down - direct) < 6.0 * _sigma(down + direct, n / 3.0)


def test_the_thermal_doubling_agrees_with_the_thermal_integral_equation_on_isotropic_scattering():
    for b, omega in ((1.0, 0.5), (3.0, 0.9)):
        f1, u1, d1 = thermal_slab(b, omega, [0.8] * 4, 1.3)
        f2, u2, d2 = thermal_doubling(b, omega, [1.0], 0.8, 1.3, 0.0)
        assert abs(f1 - f2) < 1e-9 and abs(u1 - u2) < 5e-6 and abs(d1 - d2) < 5e-6


def config4_like(tau, ssa, g, nleg, temp, sfc_temp, albedo, n=6, lam=10.0):
    case = cases.homog_lw(n=n, ext=tau / (0.1 * n), ssa=ssa, g=g, nleg=nleg, temp=temp, sfc_temp=sfc_temp, albedo=albedo, lam=lam)
    chi = np.concatenate([[1.0], np.asarray(case["components"][0]["legendre"][0], np.float64)])
    theory = thermal_doubling(tau, ssa, sampled_moments(chi, table=9001), float(planck(lam, temp)), float(planck(lam, sfc_temp)), albedo,
                              streams=HG_STREAMS)
    return case, theory


# config 4's kind of problem (Domain-Files/homogBBDomain.f95:39-66: homogeneous, isothermal 280 K over a warmer surface of
# albedo 0.1, optical depth 10, omega0 = 0.5), with a phase function whose Legendre series has converged
CONFIG4_LIKE = [(10.0, 0.5, 0.6, 48, 280.0, 300.0, 0.1), (2.0, 0.9, 0.85, 64, 250.0, 290.0, 0.4)]


@pytest.mark.parametrize("tau,ssa,g,nleg,temp,sfc,albedo", CONFIG4_LIKE)
def test_oracle_thermal_emission_forward_scattering_and_a_grey_surface(tau, ssa, g, nleg, temp, sfc, albedo):
    from oracle import oracle as O
    n = 300000
    case, (frac, up, onto) = config4_like(tau, ssa, g, nleg, temp, sfc, albedo)
    P = cases.oracle_problem(case, nsteps=9001, lw_flag=1.0)
    vw, f, _ = O.emission_weighting(P, case["temps"].transpose(2, 1, 0).reshape(-1), case["lambda_um"], case["sfc_temp"])
    assert abs(f - frac) < 2e-5
    r = O.compute_radiative_transfer(P, O.EmissionSource(vw, f), O.philox_rng(SEED, 0), n)
    assert abs(r["meanFluxUp"] - up) < 6.0 * _sigma(up, n)
    assert abs(r["meanFluxDown"] - onto) < 6.0 * _sigma(min(onto, 0.5), n) * (1.0 + albedo)


def test_the_two_deterministic_solvers_agree_on_isotropic_scattering():
    for b, omega, node in ((1.0, 1.0, 16), (2.0, 0.9, 5), (0.5, 1.0, 31)):
        mu0, up, down = doubling_slab(b, omega, [1.0], node)
        u2, d2, direct = isotropic_slab(b, omega, mu0)
        assert abs(up - u2) < 3e-6 and abs(down - d2 - direct) < 3e-6


# ---------------------------------------------------------------------------------------------------------------------
# the oracle (CPU)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("b,omega,g,nleg,node", HG_SLABS)
def test_oracle_henyey_greenstein_scattering_against_matrix_doubling(b, omega, g, nleg, node):
    """The whole angle machinery -- Legendre series -> phase function values -> inverse CDF table
    (computeInversePhaseFunction, src/inversePhaseFunctions.f95:66-174) -> computeScatteringAngle -> next_direct -- against
    a discrete-ordinates solution for the same series."""
    from oracle import oracle as O
    n = 200000
    case, chi = hg_slab(b, omega, g, nleg)
    mu0, up, down = doubling_slab(b, omega, sampled_moments(chi, table=9001), node, streams=HG_STREAMS)
    P = cases.oracle_problem(case, nsteps=9001)
    r = O.compute_radiative_transfer(P, O.solar_source(mu0, 20.0), O.philox_rng(SEED, 0), n)
    assert abs(r["meanFluxUp"] - up) < 6.0 * _sigma(up, n)
    assert abs(r["meanFluxDown"] - down) < 6.0 * _sigma(down, n)


def test_the_reference_samples_its_lobatto_polygon_not_the_series():
    """What `sampled_moments` says, measured: for the I3RC phase function the oracle (whose tables are the reference's rule
    bit for bit) follows transport theory for the polygon and is 6 sigma away from transport theory for the series itself
    (reflected flux of a conservative slab of optical depth 1 under a high sun: 0.0432 against 0.0440, 1.7 %)."""
    from oracle import oracle as O
    n = 3000000
    case, chi = hg_slab(1.0, 1.0, 0.85, 64)
    mu0, up_polygon, _ = doubling_slab(1.0, 1.0, sampled_moments(chi, table=9001), 124, streams=HG_STREAMS)
    _, up_series, _ = doubling_slab(1.0, 1.0, chi, 124, streams=HG_STREAMS)
    assert abs(sampled_moments(chi)[1] - 0.851776) < 2e-6 and abs(chi[1] - 0.85) < 1e-7
    assert abs(sampled_moments(chi, table=9001)[1] - (0.851776 - 1.0 / 9001)) < 2e-5   # (the staircase: -1 / nSteps, and see the 10^9 record)
    r = O.compute_radiative_transfer(cases.oracle_problem(case, nsteps=9001), O.solar_source(mu0, 20.0), O.philox_rng(SEED, 0), n)
    assert abs(r["meanFluxUp"] - up_polygon) < 4.0 * _sigma(up_polygon, n)
    assert abs(r["meanFluxUp"] - up_series) > 5.0 * _sigma(up_series, n)
    assert 0.012 < (up_series - up_polygon) / up_series < 0.022


def tabulated_slab(tau, ssa, nz=16):
    ang, val = cases.tabulated_two_lobe(361)
    case = slab(tau, ssa, nz=nz)
    comp = case["components"][0]
    del comp["legendre"]
    comp["tabulated"] = [(ang, val)]
    return case, (np.cos(ang.astype(np.float64))[::-1], val.astype(np.float64)[::-1])


def test_oracle_angle_value_phase_function_against_matrix_doubling():
    """The "Mie table" storage (src/scatteringPhaseFunctions.f95:104-164; inverse table from the native angles,
    inversePhaseFunctions.f95:80-88): a forward lobe plus a weak backward one on 361 angles."""
    from oracle import oracle as O
    n = 200000
    case, nodes = tabulated_slab(2.0, 0.9)
    mu0, up, down = doubling_slab(2.0, 0.9, sampled_moments(None, table=9001, nodes=nodes), 64, streams=HG_STREAMS)
    r = O.compute_radiative_transfer(cases.oracle_problem(case, nsteps=9001), O.solar_source(mu0, 20.0), O.philox_rng(SEED, 0), n)
    assert abs(r["meanFluxUp"] - up) < 6.0 * _sigma(up, n)
    assert abs(r["meanFluxDown"] - down) < 6.0 * _sigma(down, n)


@pytest.mark.parametrize("b,omega,mu0", SCATTERING_SLABS)
def test_oracle_isotropic_scattering_against_the_integral_equation(b, omega, mu0):
    from oracle import oracle as O
    n = 200000
    P = cases.oracle_problem(slab(b, omega, nz=16), nsteps=9001)
    r = O.compute_radiative_transfer(P, O.solar_source(mu0, 75.0), O.philox_rng(SEED, 0), n)
    up, down, direct = isotropic_slab(b, omega, mu0)
    # (a photon's weight is not 0 / 1 when omega < 1, and roulette adds variance: the binomial sigma is a bound from below;
    # 6 of them is the tolerance -- 0.6 % of a flux of 0.3 at this sample size)
    assert abs(r["meanFluxUp"] - up) < 6.0 * _sigma(up, n)
    assert abs(r["meanFluxDown"] - (down + direct)) < 6.0 * _sigma(down + direct, n)
    assert abs(r["meanFluxAbsorbed"] - (1.0 - up - down - direct)) < 6.0 * _sigma(up, n) + 1e-6


@pytest.mark.parametrize("tau,mu0", [(0.5, 1.0), (2.0, 0.5), (1.0, 0.2)])
def test_oracle_beer_lambert(tau, mu0):
    from oracle import oracle as O
    n = 100000
    P = cases.oracle_problem(slab(tau, 0.0), nsteps=101)
    r = O.compute_radiative_transfer(P, O.solar_source(mu0, 40.0), O.philox_rng(SEED, 0), n)
    t = float(np.exp(-tau / mu0))
    assert r["meanFluxUp"] == 0.0
    assert abs(r["meanFluxDown"] - t) < 4.5 * _sigma(t, n) + 1e-6
    assert abs(r["meanFluxAbsorbed"] - (1.0 - t)) < 4.5 * _sigma(t, n) + 1e-6


def test_oracle_conserves_energy_over_a_white_surface():
    from oracle import oracle as O
    n = 20000
    P = cases.oracle_problem(slab(1.0, 1.0, albedo=1.0), nsteps=101)
    r = O.compute_radiative_transfer(P, O.solar_source(0.6, 0.0), O.philox_rng(SEED, 0), n)
    assert abs(r["meanFluxUp"] - 1.0) < 2e-6 and abs(r["meanFluxAbsorbed"]) < 2e-6
    # (the surface sees every photon that reaches it, each time: more than the direct beam)
    assert r["meanFluxDown"] > float(np.exp(-1.0 / 0.6))


@pytest.mark.parametrize("tau", [0.25, 1.0, 3.0])
def test_oracle_isothermal_layer_over_a_black_surface(tau):
    from oracle import oracle as O
    n = 100000
    case = isothermal(tau)
    P = cases.oracle_problem(case, nsteps=101, lw_flag=1.0)
    vw, frac, _ = O.emission_weighting(P, case["temps"].transpose(2, 1, 0).reshape(-1), case["lambda_um"], case["sfc_temp"])
    assert abs(frac - 4.0 * tau / (1.0 + 4.0 * tau)) < 1e-6
    r = O.compute_radiative_transfer(P, O.EmissionSource(vw, frac), O.philox_rng(SEED, 0), n)
    e = 1.0 / (1.0 + 4.0 * tau)
    d = e * (1.0 - 2.0 * float(expn(3, tau)))
    assert abs(r["meanFluxUp"] - e) < 4.5 * _sigma(e, n)
    assert abs(r["meanFluxDown"] - d) < 4.5 * _sigma(d, n)
    assert abs(r["meanFluxAbsorbed"] + d) < 4.5 * (_sigma(e, n) + _sigma(d, n))
    # ... and level by level: the layers cool to space from the top; next to the black surface at their own temperature
    # they are nearly in balance
    prof = np.asarray(r["absorbedProfile"], np.float64)
    assert prof[-1] < 0.0 and abs(prof[0]) < abs(prof[-1])


@pytest.mark.parametrize("tau,omega,temps,sfc", THERMAL_SLABS)
def test_oracle_thermal_emission_with_scattering_against_the_integral_equation(tau, omega, temps, sfc):
    """emission_weightingNEW + newPhotonStream_BBEmission + the LW tallies, with scattering and a temperature profile."""
    from oracle import oracle as O
    n = 200000
    case, frac, up, down = thermal_case(tau, omega, temps, sfc)
    P = cases.oracle_problem(case, nsteps=101, lw_flag=1.0)
    vw, f, _ = O.emission_weighting(P, case["temps"].transpose(2, 1, 0).reshape(-1), case["lambda_um"], case["sfc_temp"])
    assert abs(f - frac) < 2e-5   # (the reference's physical constants against the tests': a few 10^-6)
    r = O.compute_radiative_transfer(P, O.EmissionSource(vw, f), O.philox_rng(SEED, 0), n)
    assert abs(r["meanFluxUp"] - up) < 6.0 * _sigma(up, n)
    assert abs(r["meanFluxDown"] - down) < 6.0 * _sigma(down, n)
    assert abs(r["meanFluxAbsorbed"] - (1.0 - up - down - frac)) < 6.0 * (_sigma(up, n) + _sigma(down, n))


@pytest.mark.parametrize("b,omega,mu0", RADIANCE_SLABS)
@pytest.mark.parametrize("rr", [False, True])
def test_oracle_radiance_of_an_isotropically_scattering_slab(b, omega, mu0, rr):
    """computeIntensityContribution (:1623-1832: the local estimate at every scattering, its transmission along the view
    ray, the normalisation) against the formal solution with the integral equation's source function -- with and without
    the roulette on the estimates (unbiased by construction, Iwabuchi 2006)."""
    from oracle import oracle as O
    case = slab(b, omega, nz=16)
    P = cases.oracle_problem(case, nsteps=9001)
    I = cases.oracle_intensity(case, RADIANCE_MUS, RADIANCE_PHIS, n_angles=1801, use_russian_roulette=rr, zeta_min=0.3)
    per, nb = 20000, 12
    runs = np.array([O.compute_radiative_transfer_intensity(P, O.solar_source(mu0, 33.0), O.philox_rng(SEED, k * per), per, I)["meanIntensity"]
                     for k in range(nb)], np.float64)
    mean, err = runs.mean(axis=0), runs.std(axis=0, ddof=1) / np.sqrt(nb)
    theory = isotropic_radiance(b, omega, mu0, RADIANCE_MUS)
    assert np.all(err < 0.02 * theory)
    assert np.all(np.abs(mean - theory) < 4.5 * err), (mean, theory, err)


@pytest.mark.parametrize("tau,omega,temps,sfc", THERMAL_SLABS[:2])
def test_oracle_thermal_radiance_against_the_formal_solution(tau, omega, temps, sfc):
    """Emission seen directly (the local estimates at launch, computeRT :510-541), attenuated, plus the scattered field:
    in the isothermal case without scattering the answer is B itself in every direction."""
    from oracle import oracle as O
    lam = 10.0
    case, _, _, _ = thermal_case(tau, omega, temps, sfc, lam)
    theory = thermal_radiance(tau, omega, planck(lam, np.asarray(temps)[::-1]), float(planck(lam, sfc)), RADIANCE_MUS)
    P = cases.oracle_problem(case, nsteps=101, lw_flag=1.0)
    vw, f, _ = O.emission_weighting(P, case["temps"].transpose(2, 1, 0).reshape(-1), lam, sfc)
    I = cases.oracle_intensity(case, RADIANCE_MUS, RADIANCE_PHIS, n_angles=181)
    per, nb = 20000, 10
    runs = np.array([O.compute_radiative_transfer_intensity(P, O.EmissionSource(vw, f), O.philox_rng(SEED, k * per), per, I)["meanIntensity"]
                     for k in range(nb)], np.float64)
    mean, err = runs.mean(axis=0), runs.std(axis=0, ddof=1) / np.sqrt(nb)
    assert np.all(err < 0.02 * theory)
    assert np.all(np.abs(mean - theory) < 4.5 * err), (mean, theory, err)


def test_oracle_radiance_of_a_forward_scattering_slab_in_the_azimuthal_mean():
    """The local estimate WITH its phase-function factor (the forward tables, tabulateForwardPhaseFunctions) against matrix
    doubling: the mean over 12 view azimuths is the azimuthal mean that the reflection operator carries."""
    from oracle import oracle as O
    case, chi = hg_slab(2.0, 0.95, 0.6, 48)
    # (theory for the SERIES: the estimate's own phase-function factor -- most of the radiance at these angles -- comes from the
    # forward tables, which hold the series; the polygon the directions are sampled from differs from it by up to 1 % at
    # single angles and by 10^-4 in this slab's fluxes, which is what the earlier scatterings contribute through)
    mu0, vmus, theory = doubling_radiance(2.0, 0.95, chi, 64, [64, 110])
    mus = [m for m in vmus for _ in range(RADIANCE_AZIMUTHS)]
    phis = [360.0 * k / RADIANCE_AZIMUTHS for _ in vmus for k in range(RADIANCE_AZIMUTHS)]
    P = cases.oracle_problem(case, nsteps=9001)
    I = cases.oracle_intensity(case, mus, phis, n_angles=1801)
    per, nb = 10000, 10
    runs = np.array([O.compute_radiative_transfer_intensity(P, O.solar_source(mu0, 0.0), O.philox_rng(SEED, k * per), per, I)["meanIntensity"]
                     for k in range(nb)], np.float64).reshape(nb, len(vmus), RADIANCE_AZIMUTHS).mean(axis=2)
    mean, err = runs.mean(axis=0), runs.std(axis=0, ddof=1) / np.sqrt(nb)
    assert np.all(err < 0.02 * theory)
    assert np.all(np.abs(mean - theory) < 4.5 * err + 0.003 * theory), (mean, theory, err)


def test_the_reference_tallies_in_single_precision_and_large_batches_show_it():
    """A property of the reference, not of the physics: its tallies are `real` (Integrators/monteCarloRadiativeTransfer.f95:102,
    `fluxUp(ix, iy) = fluxUp(ix, iy) + photonWeight` :587), so a batch that adds 10^6 weights of a non-conservative medium
    into ONE column loses 10^-3 to rounding -- the same photons in batches of 10^4 do not.  The oracle keeps the reference's
    arithmetic (and is therefore compared with theory in small batches); the product tallies in 64-bit fixed point, exactly."""
    from oracle import oracle as O
    P = cases.oracle_problem(slab(2.0, 0.9, nz=16), nsteps=1001)
    n = 1000000
    one = O.compute_radiative_transfer(P, O.solar_source(0.5, 75.0), O.philox_rng(SEED, 0), n)["meanFluxUp"]
    parts = float(np.mean([O.compute_radiative_transfer(P, O.solar_source(0.5, 75.0), O.philox_rng(SEED, k * 10000), 10000)["meanFluxUp"]
                           for k in range(n // 10000)]))
    up = isotropic_slab(2.0, 0.9, 0.5)[0]
    assert abs(parts - up) < 4.5 * _sigma(up, n)          # the same photons, small batches: theory
    assert abs(one - parts) > 4e-4                        # one large batch: rounding, not physics (seen: 1.2e-3)


# ---------------------------------------------------------------------------------------------------------------------
# the product, through the C ABI
# ---------------------------------------------------------------------------------------------------------------------
def _solar(case, mu0, phi0, n, rr=True, table=101):
    import mcbrat3d_amd as M
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=table, useRayTracing=True, useRussianRoulette=rr)
    photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 12)
    integ.resetMoments()
    assert integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, n) == n
    res = integ.reportResults()
    bad = integ.badPhotons()
    integ.finalize()
    assert bad == 0
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("tau,mu0", [(0.5, 1.0), (2.0, 0.5), (1.0, 0.2)])
def test_product_beer_lambert(tau, mu0):
    n = 1000000
    r = _solar(slab(tau, 0.0), mu0, 40.0, n)
    t = float(np.exp(-tau / mu0))
    assert r["meanFluxUp"] == 0.0
    assert abs(r["meanFluxDown"] - t) < 4.5 * _sigma(t, n) + 1e-6
    assert abs(r["meanFluxAbsorbed"] - (1.0 - t)) < 4.5 * _sigma(t, n) + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("b,omega,mu0", SCATTERING_SLABS)
def test_product_isotropic_scattering_against_the_integral_equation(b, omega, mu0):
    n = 4000000
    case = slab(b, omega, nz=16)
    import mcbrat3d_amd as M
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=True)
    photons = M.new_PhotonStream(mu0, 75.0, numberOfPhotons=10 ** 12)
    integ.resetMoments()
    assert integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, n) == n
    r = integ.reportResults()
    assert integ.badPhotons() == 0
    integ.finalize()
    up, down, direct = isotropic_slab(b, omega, mu0)
    assert abs(r["meanFluxUp"] - up) < 6.0 * _sigma(up, n)                       # (0.14 % of a flux of 0.3)
    assert abs(r["meanFluxDown"] - (down + direct)) < 6.0 * _sigma(down + direct, n)
    assert abs(r["meanFluxAbsorbed"] - (1.0 - up - down - direct)) < 6.0 * _sigma(up, n) + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("b,omega,g,nleg,node", HG_SLABS)
def test_product_henyey_greenstein_scattering_against_matrix_doubling(b, omega, g, nleg, node):
    n = 4000000
    case, chi = hg_slab(b, omega, g, nleg)
    mu0, up, down = doubling_slab(b, omega, sampled_moments(chi, table=9001), node, streams=HG_STREAMS)
    r = _solar(case, mu0, 20.0, n, table=9001)
    assert abs(r["meanFluxUp"] - up) < 6.0 * _sigma(up, n)
    assert abs(r["meanFluxDown"] - down) < 6.0 * _sigma(down, n)


@pytest.mark.gpu
def test_product_conserves_energy_over_a_white_surface():
    r = _solar(slab(1.0, 1.0, albedo=1.0), 0.6, 0.0, 200000)
    assert abs(r["meanFluxUp"] - 1.0) < 2e-6 and abs(r["meanFluxAbsorbed"]) < 2e-6
    assert r["meanFluxDown"] > float(np.exp(-1.0 / 0.6))


@pytest.mark.gpu
@pytest.mark.parametrize("tau", [0.25, 1.0, 3.0])
def test_product_isothermal_layer_over_a_black_surface(tau):
    import mcbrat3d_amd as M
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    n = 1000000
    case = isothermal(tau)
    dom = cases.product_domain(case)
    nx = len(case["xe"]) - 1
    w = M.new_Weights(nx, nx, nx)
    M.emission_weighting(dom, w, case["sfc_temp"])
    assert abs(w.fracAtmsPower - 4.0 * tau / (1.0 + 4.0 * tau)) < 1e-6
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=101, useRayTracing=True, useRussianRoulette=True, LW_flag=1.0)
    photons = M.new_PhotonStream(theseWeights=w, numberOfPhotons=10 ** 12)
    integ.resetMoments()
    assert integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, n) == n
    r = integ.reportResults()
    assert integ.badPhotons() == 0
    integ.finalize()
    e = 1.0 / (1.0 + 4.0 * tau)
    d = e * (1.0 - 2.0 * float(expn(3, tau)))
    assert abs(r["meanFluxUp"] - e) < 4.5 * _sigma(e, n)
    assert abs(r["meanFluxDown"] - d) < 4.5 * _sigma(d, n)
    assert abs(r["meanFluxAbsorbed"] + d) < 4.5 * (_sigma(e, n) + _sigma(d, n))


@pytest.mark.gpu
@pytest.mark.parametrize("tau,omega,temps,sfc", THERMAL_SLABS)
def test_product_thermal_emission_with_scattering_against_the_integral_equation(tau, omega, temps, sfc):
    import mcbrat3d_amd as M
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    n = 4000000
    case, frac, up, down = thermal_case(tau, omega, temps, sfc)
    dom = cases.product_domain(case)
    nx = len(case["xe"]) - 1
    w = M.new_Weights(nx, nx, nx)
    M.emission_weighting(dom, w, case["sfc_temp"])
    assert abs(w.fracAtmsPower - frac) < 2e-5
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=101, useRayTracing=True, useRussianRoulette=True, LW_flag=1.0)
    photons = M.new_PhotonStream(theseWeights=w, numberOfPhotons=10 ** 12)
    integ.resetMoments()
    assert integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, n) == n
    r = integ.reportResults()
    assert integ.badPhotons() == 0
    integ.finalize()
    assert abs(r["meanFluxUp"] - up) < 6.0 * _sigma(up, n)
    assert abs(r["meanFluxDown"] - down) < 6.0 * _sigma(down, n)
    assert abs(r["meanFluxAbsorbed"] - (1.0 - up - down - frac)) < 6.0 * (_sigma(up, n) + _sigma(down, n))


@pytest.mark.gpu
@pytest.mark.parametrize("b,omega,mu0", RADIANCE_SLABS)
@pytest.mark.parametrize("rr", [False, True])
def test_product_radiance_of_an_isotropically_scattering_slab(b, omega, mu0, rr):
    import mcbrat3d_amd as M
    from mcbrat3d_amd import driver
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    case = slab(b, omega, nz=16)
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=9001, minForwardTableSize=1801, intensityMus=RADIANCE_MUS, intensityPhis=RADIANCE_PHIS,
                            computeIntensity=True, useRussianRouletteForIntensity=rr)
    photons = M.new_PhotonStream(mu0, 33.0, numberOfPhotons=10 ** 12)
    integ.resetMoments()
    assert integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, 100000, 40) == 4000000
    st = driver.statistics(driver.unpack_moments(integ.moments(), dom.numX, dom.numY, dom.numZ, len(RADIANCE_MUS)))
    assert integ.badPhotons() == 0
    integ.finalize()
    mean, err = st["intensity"][0, 0, :], st["intensity_StdErr"][0, 0, :]
    theory = isotropic_radiance(b, omega, mu0, RADIANCE_MUS)
    assert np.all(err < 0.004 * theory)
    assert np.all(np.abs(mean - theory) < 4.5 * err), (mean, theory, err)


@pytest.mark.gpu
def test_product_angle_value_phase_function_against_matrix_doubling():
    n = 4000000
    case, nodes = tabulated_slab(2.0, 0.9)
    mu0, up, down = doubling_slab(2.0, 0.9, sampled_moments(None, table=9001, nodes=nodes), 64, streams=HG_STREAMS)
    r = _solar(case, mu0, 20.0, n, table=9001)
    assert abs(r["meanFluxUp"] - up) < 6.0 * _sigma(up, n)
    assert abs(r["meanFluxDown"] - down) < 6.0 * _sigma(down, n)


@pytest.mark.gpu
@pytest.mark.parametrize("albedo,mu0", [(0.0, 0.8), (0.5, 0.35)])
def test_product_layers_two_components_and_a_lambertian_surface(albedo, mu0):
    n = 4000000
    up, onto = layered_isotropic_slab(LAYERED["dtaus"], LAYERED["omegas"], mu0, albedo=albedo)
    r = _solar(layered_case(albedo), mu0, 10.0, n)
    assert abs(r["meanFluxUp"] - up) < 6.0 * _sigma(up, n)
    assert abs(r["meanFluxDown"] - onto) < 6.0 * _sigma(min(onto, 0.5), n) * (1.0 + albedo)
    assert abs(r["meanFluxAbsorbed"] - (1.0 - up - (1.0 - albedo) * onto)) < 6.0 * (_sigma(up, n) + _sigma(min(onto, 0.5), n))


@pytest.mark.gpu
@pytest.mark.parametrize("walk", ["default", "face by face", "layers", "global tallies"])
def test_product_a_homogeneous_medium_on_a_stretched_3d_grid_is_the_slab(walk):
    """... through every walk the product has for such a grid (the default plan folds a homogeneous medium into one block)."""
    import mcbrat3d_amd as M
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    n = 4000000
    case, chi = homogeneous_3d(3.0, 0.9, 0.6, 48), np.concatenate([[1.0], cases.hg_legendre(0.6, 48).astype(np.float64)])
    mu0, up, down = doubling_slab(3.0, 0.9, sampled_moments(chi, table=9001), 40, streams=HG_STREAMS)
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=True)
    tuning = {"default": {}, "face by face": dict(blockWalk=0, layerSkip=0), "layers": dict(blockWalk=0, layerSkip=1),
              "global tallies": dict(blockWalk=0, layerSkip=1, privateTallies=0)}[walk]
    if tuning:
        integ.setTuning(**tuning)
    photons = M.new_PhotonStream(mu0, 57.0, numberOfPhotons=10 ** 12)
    integ.resetMoments()
    assert integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, n) == n
    r = integ.reportResults()
    assert integ.badPhotons() == 0
    integ.finalize()
    assert abs(r["meanFluxUp"] - up) < 6.0 * _sigma(up, n)
    assert abs(r["meanFluxDown"] - down) < 6.0 * _sigma(down, n)
    assert np.all(np.abs(np.asarray(r["fluxUp"], np.float64) - up) < 6.0 * _sigma(up, n / 35.0) * 3.0)


@pytest.mark.gpu
def test_product_wide_columns_are_independent_slabs():
    n = 6000000
    taus = [2.0, 18.0, 0.5]
    r = _solar(wide_columns(taus, 0.99), 0.7, 0.0, n)
    for i, tau in enumerate(taus):
        up, down, direct = isotropic_slab(tau, 0.99, 0.7)
        assert abs(float(r["fluxUp"][i, 0]) - up) < 6.0 * _sigma(up, n / 3.0)
        assert abs(float(r["fluxDown"][i, 0]) - down - direct) < 6.0 * _sigma(down + direct, n / 3.0)


@pytest.mark.gpu
@pytest.mark.parametrize("tau,ssa,g,nleg,temp,sfc,albedo", CONFIG4_LIKE)
def test_product_thermal_emission_forward_scattering_and_a_grey_surface(tau, ssa, g, nleg, temp, sfc, albedo):
    import mcbrat3d_amd as M
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    n = 6000000
    case, (frac, up, onto) = config4_like(tau, ssa, g, nleg, temp, sfc, albedo)
    dom = cases.product_domain(case)
    nx = len(case["xe"]) - 1
    w = M.new_Weights(nx, nx, nx)
    M.emission_weighting(dom, w, case["sfc_temp"])
    assert abs(w.fracAtmsPower - frac) < 2e-5
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=9001, useRayTracing=True, useRussianRoulette=True, LW_flag=1.0)
    photons = M.new_PhotonStream(theseWeights=w, numberOfPhotons=10 ** 12)
    integ.resetMoments()
    assert integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, n) == n
    r = integ.reportResults()
    assert integ.badPhotons() == 0
    integ.finalize()
    assert abs(r["meanFluxUp"] - up) < 6.0 * _sigma(up, n)
    assert abs(r["meanFluxDown"] - onto) < 6.0 * _sigma(min(onto, 0.5), n) * (1.0 + albedo)


@pytest.mark.gpu
def test_product_radiance_of_a_forward_scattering_slab_in_the_azimuthal_mean():
    import mcbrat3d_amd as M
    from mcbrat3d_amd import driver
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    case, chi = hg_slab(2.0, 0.95, 0.6, 48)
    # (theory for the SERIES: the estimate's own phase-function factor -- most of the radiance at these angles -- comes from the
    # forward tables, which hold the series; the polygon the directions are sampled from differs from it by up to 1 % at
    # single angles and by 10^-4 in this slab's fluxes, which is what the earlier scatterings contribute through)
    mu0, vmus, theory = doubling_radiance(2.0, 0.95, chi, 64, [64, 110])
    mus = [m for m in vmus for _ in range(RADIANCE_AZIMUTHS)]
    phis = [360.0 * k / RADIANCE_AZIMUTHS for _ in vmus for k in range(RADIANCE_AZIMUTHS)]
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=9001, minForwardTableSize=1801, intensityMus=mus, intensityPhis=phis, computeIntensity=True)
    photons = M.new_PhotonStream(mu0, 0.0, numberOfPhotons=10 ** 12)
    integ.resetMoments()
    nb = 40
    assert integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, 50000, nb) == 50000 * nb
    st = driver.statistics(driver.unpack_moments(integ.moments(), dom.numX, dom.numY, dom.numZ, len(mus)))
    assert integ.badPhotons() == 0
    integ.finalize()
    per_dir, err_dir = st["intensity"][0, 0, :].reshape(len(vmus), -1), st["intensity_StdErr"][0, 0, :].reshape(len(vmus), -1)
    mean = per_dir.mean(axis=1)
    err = np.sqrt((err_dir ** 2).sum(axis=1)) / RADIANCE_AZIMUTHS * 1.5   # (the directions share their photons: correlated, so with room)
    assert np.all(err < 0.01 * theory)
    assert np.all(np.abs(mean - theory) < 4.5 * err + 0.003 * theory), (mean, theory, err)
    # ... and the field is not isotropic in azimuth (the test would pass trivially if it were)
    assert per_dir[0].max() > 1.05 * per_dir[0].min()


@pytest.mark.gpu
@pytest.mark.parametrize("tau,omega,temps,sfc", THERMAL_SLABS[:2] + [(1.0, 0.0, [285.0] * 6, 285.0)])
def test_product_thermal_radiance_against_the_formal_solution(tau, omega, temps, sfc):
    import mcbrat3d_amd as M
    from mcbrat3d_amd import driver
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    lam = 10.0
    case, _, _, _ = thermal_case(tau, omega, temps, sfc, lam)
    theory = thermal_radiance(tau, omega, planck(lam, np.asarray(temps)[::-1]), float(planck(lam, sfc)), RADIANCE_MUS)
    if omega == 0.0 and len(set(temps)) == 1 and temps[0] == sfc:  # Kirchhoff: B itself, as a share of pi B (1 + 4 tau)
        assert np.allclose(theory, 1.0 / (np.pi * (1.0 + 4.0 * tau)), rtol=2e-6)
    dom = cases.product_domain(case)
    nx = len(case["xe"]) - 1
    w = M.new_Weights(nx, nx, nx)
    M.emission_weighting(dom, w, sfc)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=101, minForwardTableSize=181, LW_flag=1.0, intensityMus=RADIANCE_MUS, intensityPhis=RADIANCE_PHIS,
                            computeIntensity=True)
    photons = M.new_PhotonStream(theseWeights=w, numberOfPhotons=10 ** 12)
    integ.resetMoments()
    assert integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, 100000, 40) == 4000000
    st = driver.statistics(driver.unpack_moments(integ.moments(), dom.numX, dom.numY, dom.numZ, len(RADIANCE_MUS)))
    assert integ.badPhotons() == 0
    integ.finalize()
    mean = st["intensity"].reshape(-1, len(RADIANCE_MUS)).mean(axis=0)
    err = np.sqrt((st["intensity_StdErr"].reshape(-1, len(RADIANCE_MUS)) ** 2).sum(axis=0)) / (nx * nx)
    assert np.all(err < 0.005 * theory)
    assert np.all(np.abs(mean - theory) < 4.5 * err), (mean, theory, err)


@pytest.mark.gpu
def test_product_tallies_do_not_depend_on_the_batch_size():
    """... where the reference's single-precision tallies do (test_the_reference_tallies_in_single_precision_...): the same
    2x10^6 photons in one batch and in 200, to the last digit of the float the means are reported in."""
    import mcbrat3d_amd as M
    from mcbrat3d_amd import driver
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    case = slab(2.0, 0.9, nz=16)
    dom = cases.product_domain(case)
    means = []
    for ppb, nb in ((2000000, 1), (10000, 200)):
        integ = M.new_Integrator(dom)
        integ.specifyParameters(minInverseTableSize=1001, useRayTracing=True, useRussianRoulette=True)
        photons = M.new_PhotonStream(0.5, 75.0, numberOfPhotons=10 ** 12)
        integ.resetMoments()
        assert integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, ppb, nb) == ppb * nb
        st = driver.statistics(driver.unpack_moments(integ.moments(), dom.numX, dom.numY, dom.numZ))
        means.append((float(st["meanFluxUp"]), float(st["meanFluxDown"]), float(st["meanFluxAbsorbed"])))
        integ.finalize()
    assert np.allclose(means[0], means[1], rtol=0, atol=2e-7), means
    up = isotropic_slab(2.0, 0.9, 0.5)[0]
    assert abs(means[0][0] - up) < 4.5 * _sigma(up, 2000000)
