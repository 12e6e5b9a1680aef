"""Spectrally integrated ("broadband") runs: the loop over wavelength domains that surrounds the
hot path in Drivers/monteCarloDriver.f95 -- emitted power per wavelength (:304-399), its running
CDF (:417-433), the split of the photons over wavelengths (getFrequencyDistr,
src/emissionAndBroadBandWeights.f95:552-572, driver :438-449) and the worker loop over
(wavelength, batch) work units (:889-1085).  Every photon carries the same weight; the moment
arrays simply keep accumulating over the wavelength domains, and the spectrally integrated flux
scales the result (solarFlux, :1188-1228).

"kDistribution" in BASELINE.json's config 4 has no code behind it in the reference
(src/kDistribution.f95 is a set of empty stubs): spectral integration IS this loop."""
import numpy as np

from ._capi import McbratError
from .illumination import emission_weighting, new_PhotonStream, new_Weights


def spectral_widths(lambdas):
    """dLambda per wavelength as the driver forms it (:325-360): half the distance between the
    neighbours, one-sided at both ends."""
    lam = np.asarray(lambdas, np.float64)
    n = len(lam)
    if n == 1:
        return np.ones(1)
    d = np.empty(n)
    d[0] = abs(lam[1] - lam[0])
    d[1:-1] = np.abs((lam[2:] - lam[:-2]) / 2.0)
    d[-1] = abs(lam[-1] - lam[-2])
    return d


def emitted_flux_cdf(fluxes):
    """Compensated running sum, normalised, last element pinned to 1 (:422-433).
    Returns (cdf, total flux)."""
    c = np.array(fluxes, np.float64)
    corr = 0.0
    for i in range(1, len(c)):
        contrib = c[i] - corr
        s = c[i - 1] + contrib
        corr = (s - c[i - 1]) - contrib
        c[i] = s
    total = c[-1]
    c = c / total
    c[-1] = 1.0
    return c, total


def frequency_distribution(cdf, totalPhotons, seed=10):
    """getFrequencyDistr: the reference draws one uniform per photon from its MT stream (on every rank) and counts
    the photons that fall into each interval of the power CDF (smallest i with U <= CDF(i), findCDFIndex) -- an
    O(numPhotons) host loop, 10^9 draws for a production run.  The counts are one sample of
    Multinomial(totalPhotons, diff(CDF)); here that sample is drawn directly (conditional binomials, O(numLambda))
    from a counter-based generator keyed by the seed."""
    cdf = np.asarray(cdf, np.float64)
    p = np.diff(np.concatenate([[0.0], cdf]))
    p = np.clip(p, 0.0, None)
    rng = np.random.Generator(np.random.Philox(key=int(seed)))
    return rng.multinomial(int(totalPhotons), p / p.sum()).astype(np.int64)


def run_thermal(integrator, domains, surfaceTemp, numPhotonsPerBatch, numBatches, randomNumbers, seed=10):
    """Thermal (LW_flag >= 0) broadband run over `domains` (one Domain per wavelength, same grid,
    domain.lambda_um set).  Returns (photon counts per wavelength, spectrally integrated flux).
    The integrator's moment arrays hold the result (driver.statistics(..., solarFlux=flux))."""
    if not domains:
        raise McbratError("run_thermal: no wavelength domains")
    widths = spectral_widths([d.lambda_um for d in domains])
    weights, fluxes = [], []
    for dom, dl in zip(domains, widths):  # set-up pass :304-399
        w = new_Weights(dom.numX, dom.numY, dom.numZ)
        fluxes.append(emission_weighting(dom, w, surfaceTemp, dLambda=dl))
        weights.append(w)
    cdf, total_flux = emitted_flux_cdf(fluxes)
    counts = frequency_distribution(cdf, int(numPhotonsPerBatch) * int(numBatches), seed)
    integrator.specifyParameters(LW_flag=1.0)
    for dom, w, n in zip(domains, weights, counts):  # worker loop :889-1085
        left = int(n)
        if left == 0:
            continue
        photons = new_PhotonStream(theseWeights=w, numberOfPhotons=left)
        full, rest = divmod(left, int(numPhotonsPerBatch))
        if full:
            integrator.computeRadiativeTransfer(dom, randomNumbers, photons, int(numPhotonsPerBatch), full)
        if rest:
            integrator.computeRadiativeTransfer(dom, randomNumbers, photons, rest, 1)
    return counts, total_flux
