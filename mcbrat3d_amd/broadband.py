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


def solar_weighting(solarSourceFunction, lambdas, solarMu, spectrRespFunc=None):
    """solar_Weighting (src/emissionAndBroadBandWeights.f95:149-217): power per wavelength = dLambda x solarMu x source
    function [x instrument response], dLambda as spectral_widths(), summed with the same compensated running sum,
    normalised.  Returns (totalPowerCDF, spectrally integrated flux)."""
    src = np.asarray(solarSourceFunction, np.float64)
    lam = np.asarray(lambdas, np.float64)
    if src.size != lam.size or lam.size < 1:
        raise McbratError("solar_Weighting: source function and wavelengths must have the same length")
    # (solarMu is a default real in the reference; a single wavelength gets dLambda = 1 as in the thermal branch,
    #  driver :325-360 -- the reference's own loop would read lambdas(2) there)
    power = spectral_widths(lam) * float(np.float32(solarMu)) * src
    if spectrRespFunc is not None:
        power = power * np.asarray(spectrRespFunc, np.float64)
    return emitted_flux_cdf(power)


def device_frequency_distribution(integrator, cdf, totalPhotons, seed=10, firstDraw=0):
    """getFrequencyDistr with the draws made and counted on the GPU (mcbrat_frequency_distribution): one uniform per
    photon against the power CDF, as the reference, without its O(numPhotons) host loop."""
    return integrator.frequencyDistribution(cdf, totalPhotons, seed, firstDraw)


class SpectralRun:
    """A spectrally integrated run with every wavelength's optical properties RESIDENT on the device: one integrator
    (context) per wavelength domain, uploaded once by prepare(), all accumulating into one moment array -- the
    reference re-reads and re-expands the domain of a wavelength for every work unit (read_SSPTable inside the worker
    loop, Drivers/monteCarloDriver.f95:936).  The loop over (wavelength, batch) units (:889-1085) then only launches
    kernels.  The photons are split over wavelengths on the device (getFrequencyDistr, driver :438-449)."""

    def __init__(self, M, domains, device=0, overlap=True, **parameters):
        """overlap: the wavelengths' tracing kernels may overlap on the GPU (each integrator asynchronous, the finish chains
        of consecutive calls ordered on the device by chainAfter): the tail of one wavelength's launch is filled by the next
        wavelength's photons.  Bitwise the results of overlap=False (every call waited for by the host)."""
        if not domains:
            raise McbratError("SpectralRun: no wavelength domains")
        self.overlap = bool(overlap)
        self.M, self.domains = M, list(domains)
        self.integrators = [M.new_Integrator(d, device=device) for d in self.domains]
        for it in self.integrators:
            it.specifyParameters(**parameters)
        self.first = self.integrators[0]
        self._bound, self._bound_ptr, self._last = False, None, None
        self.streams, self.counts, self.cdf, self.totalFlux = None, None, None, 0.0

    def _bind(self):
        # The first integrator's moment array moves when its length changes (set_grid; specifyParameters with another
        # number of intensity directions frees and re-allocates it): the others are bound again whenever the address
        # they hold is no longer the one in use.  Contexts that share an array are only ordered in synchronous mode.
        ptr = self.first.momentsDevicePointer()
        if not self._bound or ptr != self._bound_ptr:
            for it in self.integrators:
                it._shares_moments = len(self.integrators) > 1
                it.setAsync(self.overlap and len(self.integrators) > 1, _chained=True)
            for it in self.integrators[1:]:
                it.bindMoments(ptr)
            self._bound, self._bound_ptr = True, ptr

    def bindMoments(self, device_ptr):
        """All wavelengths accumulate into a caller-owned device buffer (the tensor a multi-GPU run all-reduces)."""
        self.first.bindMoments(device_ptr)
        self._bound = False
        self._bind()

    def prepare_thermal(self, surfaceTemp):
        """Set-up pass (driver :304-433): emission weights and emitted power of every wavelength, the power CDF; uploads."""
        widths = spectral_widths([d.lambda_um for d in self.domains])
        self.streams, fluxes = [], []
        for dom, it, dl in zip(self.domains, self.integrators, widths):
            w = new_Weights(dom.numX, dom.numY, dom.numZ)
            fluxes.append(emission_weighting(dom, w, surfaceTemp, dLambda=dl))
            it.specifyParameters(LW_flag=1.0)
            ps = new_PhotonStream(theseWeights=w, numberOfPhotons=10 ** 15)
            it.prepare(dom, ps)
            self.streams.append(ps)
        self.cdf, self.totalFlux = emitted_flux_cdf(fluxes)
        self._bind()
        return self.totalFlux

    def prepare_solar(self, solarMu, solarAzimuth, solarSourceFunction, lambdas=None, spectrRespFunc=None):
        """Set-up pass of a solar run (driver :452-505): solar_Weighting; uploads."""
        lam = [d.lambda_um for d in self.domains] if lambdas is None else lambdas
        self.cdf, self.totalFlux = solar_weighting(solarSourceFunction, lam, solarMu, spectrRespFunc)
        self.streams = []
        for dom, it in zip(self.domains, self.integrators):
            it.specifyParameters(LW_flag=-1.0)
            ps = new_PhotonStream(solarMu, solarAzimuth, numberOfPhotons=10 ** 15)
            it.prepare(dom, ps)
            self.streams.append(ps)
        self._bind()
        return self.totalFlux

    def run(self, numPhotonsPerBatch, numBatches, randomNumbers, seed=10, counts=None):
        """The worker loop: photons per wavelength from the power CDF (on the device unless `counts` is given), then
        for every wavelength its photons in batches of numPhotonsPerBatch (+ one smaller batch for the rest).  No host
        data is touched.  Returns the counts; the moments (shared) are read from any integrator."""
        if self.streams is None:
            raise McbratError("SpectralRun: call prepare_thermal or prepare_solar first")
        self._bind()
        total = int(numPhotonsPerBatch) * int(numBatches)
        self.counts = np.asarray(counts, np.int64) if counts is not None else \
            device_frequency_distribution(self.first, self.cdf, total, seed)
        last = self._last
        for dom, it, ps, n in zip(self.domains, self.integrators, self.streams, self.counts):
            full, rest = divmod(int(n), int(numPhotonsPerBatch))
            ps.currentPhoton = 1
            for ppb, nb in ((int(numPhotonsPerBatch), full), (rest, 1 if rest else 0)):
                if nb:
                    if self.overlap and last is not None and last is not it:
                        it.chainAfter(last)  # (the device orders the two contexts' finish kernels; the host does not wait)
                    it.computeRadiativeTransfer(dom, randomNumbers, ps, ppb, nb)
                    last = it
        self._last = last
        return self.counts

    def set_overlap(self, overlap):
        """Switch between overlapping and host-serialised wavelengths (takes effect with the next run)."""
        for it in self.integrators:
            it.synchronize()
        self.overlap = bool(overlap)
        self._bound = False
        self._bind()

    def resetMoments(self):
        self._bind()
        for it in self.integrators:  # (everything enqueued so far has written the array that is about to be cleared)
            it.synchronize()
        self.first.resetMoments()
        self.first.synchronize()
        self._last = self.first

    def moments(self):
        for it in self.integrators:
            it.synchronize()
        return self.first.moments()

    def finalize(self):
        for it in self.integrators[1:]:
            it.finalize()
        self.first.finalize()


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
