"""The integrator: host-side mirror of module monteCarloRadiativeTransfer
(Integrators/monteCarloRadiativeTransfer.f95, public list :121-123) over the C ABI.

    new_Integrator(domain)                    :129-201
    specifyParameters(...)                    :1046-1484
    computeRadiativeTransfer(...)             :209-391  (photon loop computeRT :393-841 on the GPU)
    reportResults(...)                        :845-1042
    finalize_Integrator                       :1486-1547

Failures raise McbratError with the message the reference would push on its
ErrorMessage stack; there is no CPU path."""
import ctypes as C

import numpy as np

from ._capi import Counters, FATE_DTYPE, McbratError, check, lib, ptr

defaultMinInverseTableSize = 9001  # :24-25
defaultMinForwardTableSize = 9001
defaultHybridPhaseFunWidth, maxHybridPhaseFunWidth = 7.0, 30.0  # :26-27
defaultZetaMin = 0.3  # :29


class RandomNumberSequence:
    """Stands where the reference passes type(randomNumberSequence): the Philox key and the
    id of the next photon.  Photon ids, not generator state, carry the stream, so any split
    of a run over GPUs or batches reproduces the same photons."""

    def __init__(self, seed=10, firstPhotonId=0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.nextPhotonId = int(firstPhotonId)


def new_RandomNumberSequence(seed=10, firstPhotonId=0):
    if not np.isscalar(seed):  # (/ iseed, thisProc, thisThread /), monteCarloDriver.f95:901
        s = [int(x) & 0xFFFFFFFF for x in seed]
        seed = s[0] | ((s[1] if len(s) > 1 else 0) << 32)
        seed ^= (s[2] if len(s) > 2 else 0) << 48
    return RandomNumberSequence(seed, firstPhotonId)


class Integrator:
    def __init__(self, atmosphere, device=0):
        """new_Integrator(atmosphere, status)."""
        self._lib = lib()
        self._ctx = self._lib.mcbrat_create(int(device))
        if not self._ctx:
            raise McbratError("new_Integrator: no usable HIP device %d (the MI355X library has no CPU fallback)" % device)
        self.device = int(device)
        self._atmosphere = atmosphere
        self.useSurfaceBDRF = False
        self.surfaceBDRF = None
        self.minInverseTableSize = defaultMinInverseTableSize
        self.useRayTracing = True
        self.useRussianRoulette = True
        self.LW_flag = -1.0
        # radiance by local estimation (:72-97)
        self.computeIntensity = False
        self.intensityMus = np.zeros(0, np.float32)
        self.intensityPhis = np.zeros(0, np.float32)
        self.minForwardTableSize = defaultMinForwardTableSize
        self.useHybridPhaseFunsForIntenCalcs = False
        self.hybridPhaseFunWidth = defaultHybridPhaseFunWidth
        self.numOrdersOrigPhaseFunIntenCalcs = 0
        self.useRussianRouletteForIntensity = False
        self.zetaMin = defaultZetaMin
        self.limitIntensityContributions = False
        self.maxIntensityContribution = float(np.finfo(np.float32).max)
        self._param_token = None
        self._intensity_token = None
        self._domain_token = None
        self._source_token = None
        self._loaded_domain = None   # strong references: what the device buffers were filled from
        self._loaded_weights = None
        self._dims = None
        self._load_grid(atmosphere)
        self.readyToCompute = True

    # -- life cycle -----------------------------------------------------------------
    def finalize(self):
        if getattr(self, "_ctx", None):
            self._lib.mcbrat_destroy(self._ctx)
            self._ctx = None
        self.readyToCompute = False

    def __del__(self):
        try:
            self.finalize()
        except Exception:
            pass

    def isReady_Integrator(self):
        return bool(self.readyToCompute)

    def copy_Integrator(self):
        """copy_Integrator (:1296-1376): a second integrator on the same domain with the same parameters.  The
        reference also copies the last batch's results; here results live in the context that computed them."""
        new = Integrator(self._atmosphere, self.device)
        new.specifyParameters(minInverseTableSize=self.minInverseTableSize, useRayTracing=self.useRayTracing,
                              useRussianRoulette=self.useRussianRoulette, LW_flag=self.LW_flag,
                              intensityMus=self.intensityMus, intensityPhis=self.intensityPhis,
                              computeIntensity=self.computeIntensity, minForwardTableSize=self.minForwardTableSize,
                              useRussianRouletteForIntensity=self.useRussianRouletteForIntensity, zetaMin=self.zetaMin,
                              useHybridPhaseFunsForIntenCalcs=self.useHybridPhaseFunsForIntenCalcs,
                              hybridPhaseFunWidth=self.hybridPhaseFunWidth,
                              numOrdersOrigPhaseFunIntenCalcs=self.numOrdersOrigPhaseFunIntenCalcs,
                              limitIntensityContributions=self.limitIntensityContributions,
                              maxIntensityContribution=self.maxIntensityContribution, surfaceBDRF=self.surfaceBDRF)
        return new

    def _check(self, rc):
        check(self._ctx, rc)

    def _load_grid(self, dom):
        info = dom.getInfo_Domain()
        self._check(self._lib.mcbrat_set_grid(self._ctx, info["numX"], info["numY"], info["numZ"],
                                              ptr(info["xPosition"]), ptr(info["yPosition"]), ptr(info["zPosition"])))
        self._dims = (info["numX"], info["numY"], info["numZ"])
        self._domain_token = None
        self._source_token = None

    # -- specifyParameters ------------------------------------------------------------
    def specifyParameters(self, minInverseTableSize=None, useRayTracing=None, useRussianRoulette=None, LW_flag=None,
                          computeIntensity=None, intensityMus=None, intensityPhis=None, minForwardTableSize=None,
                          useRussianRouletteForIntensity=None, zetaMin=None, useHybridPhaseFunsForIntenCalcs=None,
                          hybridPhaseFunWidth=None, numOrdersOrigPhaseFunIntenCalcs=None,
                          limitIntensityContributions=None, maxIntensityContribution=None, surfaceBDRF=None,
                          **unsupported):
        # intensity keywords, :1130-1160 and :1186-1283
        if (intensityMus is None) != (intensityPhis is None):
            raise McbratError("specifyParameters: Both or neither of intensityMus, intensityPhis must be supplied")
        if intensityMus is not None:
            mus = np.ascontiguousarray(intensityMus, np.float32).reshape(-1)
            phis = np.ascontiguousarray(intensityPhis, np.float32).reshape(-1)
            if mus.size != phis.size:
                raise McbratError("specifyParameters: intensityMus, intensityPhis must be the same length")
            self.intensityMus, self.intensityPhis = mus, phis
            self.computeIntensity = mus.size > 0
        if computeIntensity is not None:
            if computeIntensity and self.intensityMus.size == 0:
                raise McbratError("specifyParameters: Can't compute intensity without specifying directions.")
            if not computeIntensity and intensityMus is None:
                self.computeIntensity = False
        if minForwardTableSize is not None:
            self.minForwardTableSize = max(int(minForwardTableSize), defaultMinForwardTableSize)
            self._domain_token = None
        if useRussianRouletteForIntensity is not None:
            self.useRussianRouletteForIntensity = bool(useRussianRouletteForIntensity)
        if zetaMin is not None and zetaMin >= 0.0:  # :1194-1201 (a negative value is ignored with a warning)
            self.zetaMin = float(zetaMin)
        if useHybridPhaseFunsForIntenCalcs is not None:
            self.useHybridPhaseFunsForIntenCalcs = bool(useHybridPhaseFunsForIntenCalcs)
            self._domain_token = None
        if hybridPhaseFunWidth is not None:  # :1108-1114, :1209-1215
            if hybridPhaseFunWidth > maxHybridPhaseFunWidth or hybridPhaseFunWidth < 0.0:
                raise McbratError("specifyParameters: hybridPhaseFunWidth out of range (0 to 30degrees).")
            self.hybridPhaseFunWidth = float(hybridPhaseFunWidth) if 0 < hybridPhaseFunWidth < maxHybridPhaseFunWidth \
                else defaultHybridPhaseFunWidth
            self._domain_token = None
        if numOrdersOrigPhaseFunIntenCalcs is not None:
            if numOrdersOrigPhaseFunIntenCalcs < 0:
                raise McbratError("specifyParameters: numOrdersOrigPhaseFunIntenCalcs must be >= 0")
            self.numOrdersOrigPhaseFunIntenCalcs = int(numOrdersOrigPhaseFunIntenCalcs)
        if limitIntensityContributions is not None:
            self.limitIntensityContributions = bool(limitIntensityContributions)
        if maxIntensityContribution is not None and maxIntensityContribution > 0.0:
            self.maxIntensityContribution = float(maxIntensityContribution)
        if surfaceBDRF is not None:  # :1091-1094, :1173-1176
            if not (hasattr(surfaceBDRF, "isReady_surfaceDescription") and surfaceBDRF.isReady_surfaceDescription()):
                raise McbratError("specifyParameters: surface description isn't valid.")
            x, y = surfaceBDRF.xPosition, surfaceBDRF.yPosition
            refl = np.ascontiguousarray(surfaceBDRF.BRDFParameters[0].T, np.float32)  # x fastest
            self._check(self._lib.mcbrat_set_surface_description(self._ctx, int(x.size), int(y.size), ptr(x), ptr(y), ptr(refl)))
            self.useSurfaceBDRF = True
            self.surfaceBDRF = surfaceBDRF
        for k, v in unsupported.items():
            if v not in (None, False):
                raise McbratError("specifyParameters: keyword %s is not supported by the MI355X integrator" % k)
        if minInverseTableSize is not None:  # :1103-1106, :1183-1184 (smaller values are ignored)
            self.minInverseTableSize = max(int(minInverseTableSize), defaultMinInverseTableSize)
            self._domain_token = None
        if useRayTracing is not None:
            self.useRayTracing = bool(useRayTracing)
        if useRussianRoulette is not None:
            self.useRussianRoulette = bool(useRussianRoulette)
        if LW_flag is not None:
            self.LW_flag = float(LW_flag)
        self._push_parameters()

    def _push_parameters(self):
        # (what the library holds is remembered by content: a step of a small domain is two milliseconds, and the host side
        # of a call is part of it)
        token = (self.useRayTracing, self.useRussianRoulette, self.LW_flag)
        if token != self._param_token:
            self._check(self._lib.mcbrat_specify_parameters(self._ctx, int(self.useRayTracing),
                                                            int(self.useRussianRoulette), C.c_float(self.LW_flag)))
            self._param_token = token
        self._push_intensity()

    def _push_intensity(self):
        n = int(self.intensityMus.size) if self.computeIntensity else 0
        token = (n, self.intensityMus.tobytes(), self.intensityPhis.tobytes(), self.useRussianRouletteForIntensity, self.zetaMin,
                 self.useHybridPhaseFunsForIntenCalcs, self.numOrdersOrigPhaseFunIntenCalcs, self.limitIntensityContributions,
                 self.maxIntensityContribution)
        if token == self._intensity_token:
            return
        self._check(self._lib.mcbrat_specify_intensity(
            self._ctx, n, ptr(self.intensityMus) if n else None, ptr(self.intensityPhis) if n else None,
            int(self.useRussianRouletteForIntensity), C.c_float(self.zetaMin), int(self.useHybridPhaseFunsForIntenCalcs),
            int(self.numOrdersOrigPhaseFunIntenCalcs), int(self.limitIntensityContributions),
            C.c_float(self.maxIntensityContribution)))
        self._intensity_token = token
        self._domain_token = None  # forward tables go with the optics

    def numIntensityDirections(self):
        return int(self.intensityMus.size) if self.computeIntensity else 0

    def setTuning(self, blocksPerCU=-1, eventThreshold=0, maxBatchesInFlight=-1, privateTallies=-1, blockSize=-1,
                  launchThreshold=0, surfaceThreshold=0, brickLayout=-1, layerSkip=-1, blockWalk=-1):
        self._check(self._lib.mcbrat_set_tuning(self._ctx, blocksPerCU, eventThreshold, maxBatchesInFlight,
                                                privateTallies, blockSize, launchThreshold, surfaceThreshold, brickLayout))
        if layerSkip >= 0 or blockWalk >= 0:
            self._check(self._lib.mcbrat_set_walk_options(self._ctx, int(layerSkip), int(blockWalk)))

    def setOption(self, **options):
        """Scheduling options by name (include/mcbrat.h: mcbrat_set_option), e.g. jumpThreshold=16."""
        for name, value in options.items():
            self._check(self._lib.mcbrat_set_option(self._ctx, name.encode(), int(value)))

    def walkMode(self):
        """The walk a flux run of the loaded domain uses (decided by the library's launch plan, mcbrat_get_walk_mode)."""
        m = int(self._lib.mcbrat_get_walk_mode(self._ctx))
        return {"layerSkip": bool(m & 1), "blockWalk": bool(m & 2), "clearAirFlight": bool(m & 4),
                # tallies private to a workgroup in LDS; the wide plan: one workgroup of 1024 lanes per compute unit owns its LDS;
                # the block walk with the per-cell optics left in global memory (extinction per block in LDS)
                "privateTallies": bool(m & 64), "widePlan": bool(m & 16), "opticsInGlobalMemory": bool(m & 32),
                "opticsPerBlock": bool(m & 128), "emissionCdfTopInLds": bool(m & 256)}

    def badPhotons(self):
        """Photons dropped by a loop bound of the kernels since this integrator was created (the reference's nBad, :562-563)."""
        return self.counters()["badPhotons"]

    def firstDrop(self):
        """What a loop bound dropped first since this integrator was created (which bound, photon id, state, cell, direction);
        "" while badPhotons() is 0."""
        msg = self._lib.mcbrat_first_drop(self._ctx)
        return msg.decode() if msg else ""

    def eventThreshold(self):
        return int(self._lib.mcbrat_get_event_threshold(self._ctx))

    def setAsync(self, enable=True, _chained=False):
        """Let consecutive computeRadiativeTransfer / resetMoments calls overlap on the GPU (include/mcbrat.h)."""
        if enable and getattr(self, "_shares_moments", False) and not _chained:
            raise McbratError("setAsync: this integrator accumulates into a moment array it shares with other integrators "
                              "(SpectralRun); their finish kernels are only ordered in synchronous mode, or by chainAfter")
        self._check(self._lib.mcbrat_set_async(self._ctx, int(bool(enable))))

    def chainAfter(self, previous):
        """The finish kernels of this integrator's next call follow everything `previous` (an integrator that accumulates
        into the same moment array) has enqueued so far (mcbrat_chain_after)."""
        self._check(self._lib.mcbrat_chain_after(self._ctx, previous._ctx))

    def synchronize(self):
        self._check(self._lib.mcbrat_synchronize(self._ctx))

    def streamWaitDone(self, hip_stream):
        """Make a caller's HIP stream (raw handle, e.g. torch.cuda.current_stream().cuda_stream) wait for the work enqueued so far."""
        self._check(self._lib.mcbrat_stream_wait_done(self._ctx, C.c_void_p(hip_stream)))

    def waitStream(self, hip_stream):
        """Make the next write to the moments wait for what the caller's HIP stream has enqueued so far."""
        self._check(self._lib.mcbrat_wait_stream(self._ctx, C.c_void_p(hip_stream)))

    # -- computeRadiativeTransfer -------------------------------------------------------
    def _load_domain(self, dom):
        # What is on the device is remembered by CONTENT, never by object identity (CPython reuses the ids of freed
        # objects): a strong reference to the domain, its version counter (addOpticalComponent and the albedo setter
        # bump it) and the parameters the tables depend on.
        token = (dom._version, self.minInverseTableSize, self.numIntensityDirections() > 0,
                 self.minForwardTableSize, self.useHybridPhaseFunsForIntenCalcs, self.hybridPhaseFunWidth)
        if dom is self._loaded_domain and token == self._domain_token:
            return
        info = dom.getInfo_Domain()  # (expands the component arrays if they are stale)
        token = (dom._version,) + token[1:]
        if (info["numX"], info["numY"], info["numZ"]) != self._dims:
            raise McbratError("computeRadiativeTransfer: domain does not match the integrator's grid")
        nc = info["numberOfComponents"]
        tot = np.ascontiguousarray(info["totalExt"], np.float64).reshape(-1)
        cum = np.ascontiguousarray(info["cumExt"], np.float64).reshape(-1)
        ssa = np.ascontiguousarray(info["ssa"], np.float64).reshape(-1)
        pfi = np.ascontiguousarray(info["phaseFuncI"], np.int32).reshape(-1)
        self._check(self._lib.mcbrat_set_optics(self._ctx, nc, ptr(tot), ptr(cum), ptr(ssa), ptr(pfi), info["albedo"]))
        tables = dom.tabulateInversePhaseFunctions(self.minInverseTableSize)  # :280
        for c, t in enumerate(tables):
            t = np.ascontiguousarray(t, np.float32)
            self._check(self._lib.mcbrat_set_inverse_table(self._ctx, c + 1, t.shape[1], t.shape[0], ptr(t)))
        if self.numIntensityDirections() > 0:  # :281-285 forward tables only when intensity is computed
            tab, orig = dom.tabulateForwardPhaseFunctions(self.minForwardTableSize, self.useHybridPhaseFunsForIntenCalcs,
                                                          self.hybridPhaseFunWidth)
            for c, (t, o) in enumerate(zip(tab, orig)):
                t = np.ascontiguousarray(t, np.float32)
                o = np.ascontiguousarray(o, np.float32)
                self._check(self._lib.mcbrat_set_forward_table(self._ctx, c + 1, t.shape[1], t.shape[0], ptr(t),
                                                               ptr(o) if self.useHybridPhaseFunsForIntenCalcs else None))
        self._loaded_domain = dom
        self._domain_token = token
        self._source_token = None
        self._loaded_weights = None

    def _load_source(self, photons):
        if photons.kind == "Directional":  # the geometry itself is the token
            token = ("Directional", float(photons.solarMu), float(photons.solarAzimuth))
        else:  # strong reference to the weights + their version (emission_weighting bumps it)
            token = ("BBEmission", photons.weights._version, float(photons.weights.fracAtmsPower))
            if photons.weights is not self._loaded_weights:
                self._source_token = None
        if token == self._source_token:
            return
        if photons.kind == "Directional":
            self._check(self._lib.mcbrat_set_source_solar(self._ctx, C.c_float(photons.solarMu),
                                                          C.c_float(photons.solarAzimuth)))
        else:
            self._check(self._lib.mcbrat_set_source_emission(self._ctx, ptr(photons.weights.voxelWeights),
                                                             photons.weights.fracAtmsPower))
            self._loaded_weights = photons.weights
        self._source_token = token

    def prepare(self, thisDomain, incomingPhotons):
        """Everything computeRadiativeTransfer would upload for this domain and source -- optical grids, inverse (and
        forward) tables, emission CDF -- now, so that later calls with the same objects trace without touching the host
        data again (spectrally integrated runs keep one prepared integrator per wavelength)."""
        self.specifyParameters()
        self._load_domain(thisDomain)
        self._load_source(incomingPhotons)

    def computeRadiativeTransfer(self, thisDomain, randomNumbers, incomingPhotons, numPhotonsPerBatch, numBatches=1):
        """Traces numBatches batches of numPhotonsPerBatch photons (the reference call is
        numBatches = 1).  Returns numPhotonsProcessed.  reportResults() then returns the
        LAST batch, and the moment arrays hold every batch (see moments())."""
        if not self.readyToCompute:
            raise McbratError("computeRadiativeTransfer: problem not completely specified.")
        n = min(int(numPhotonsPerBatch), incomingPhotons.numberOfPhotons - incomingPhotons.currentPhoton + 1) \
            if numBatches == 1 else int(numPhotonsPerBatch)
        if n < 1:
            raise McbratError("computeRadiativeTransfer: Didn't process any photons.")
        self._push_parameters()  # push current flags
        self._load_domain(thisDomain)
        self._load_source(incomingPhotons)
        done = C.c_int64(0)
        self._check(self._lib.mcbrat_compute_radiative_transfer(self._ctx, randomNumbers.seed, randomNumbers.nextPhotonId,
                                                                n, int(numBatches), C.byref(done)))
        randomNumbers.nextPhotonId += done.value
        incomingPhotons.currentPhoton += done.value
        return done.value

    # -- reportResults ----------------------------------------------------------------
    def reportResults(self):
        nx, ny, nz = self._dims
        mu, md, ma = C.c_float(), C.c_float(), C.c_float()
        up, dn, ab = (np.zeros(nx * ny, np.float32) for _ in range(3))
        prof = np.zeros(nz, np.float32)
        vol = np.zeros(nx * ny * nz, np.float32)
        self._check(self._lib.mcbrat_report_results(self._ctx, C.addressof(mu), C.addressof(md), C.addressof(ma), ptr(up),
                                                    ptr(dn), ptr(ab), ptr(prof), ptr(vol)))
        f2 = lambda a: a.reshape(ny, nx).T  # noqa: E731  -> [ix, iy]
        res = dict(meanFluxUp=mu.value, meanFluxDown=md.value, meanFluxAbsorbed=ma.value,
                   fluxUp=f2(up), fluxDown=f2(dn), fluxAbsorbed=f2(ab), absorbedProfile=prof,
                   volumeAbsorption=vol.reshape(nz, ny, nx).transpose(2, 1, 0))
        nd = self.numIntensityDirections()
        if nd > 0:  # meanIntensity(direction), intensity(x, y, direction) :980-1010
            mean_i = np.zeros(nd, np.float32)
            inten = np.zeros(nd * nx * ny, np.float32)
            self._check(self._lib.mcbrat_report_intensity(self._ctx, ptr(mean_i), ptr(inten)))
            res.update(meanIntensity=mean_i, intensity=inten.reshape(nd, ny, nx).transpose(2, 1, 0))
        return res

    # -- batch moments (what the driver keeps in *Stats and reduces over processes) -----
    def momentsLength(self):
        return int(self._lib.mcbrat_moments_length(self._ctx))

    def bindMoments(self, device_ptr):
        self._check(self._lib.mcbrat_bind_moments(self._ctx, C.c_void_p(device_ptr)))

    def momentsDevicePointer(self):
        """Device address of the moment array in use (to be bound by the integrators of the other wavelengths)."""
        ptr_ = self._lib.mcbrat_moments_device_pointer(self._ctx)
        if not ptr_:
            raise McbratError("moments: no moment array (grid not set)")
        return int(ptr_)

    def frequencyDistribution(self, cdf, totalPhotons, seed=10, firstDraw=0):
        """getFrequencyDistr on the device: photons per wavelength (one uniform per photon against the power CDF)."""
        cdf = np.ascontiguousarray(cdf, np.float64)
        out = np.zeros(cdf.size, np.int64)
        self._check(self._lib.mcbrat_frequency_distribution(self._ctx, int(seed) & 0xFFFFFFFFFFFFFFFF, int(firstDraw), int(cdf.size),
                                                            ptr(cdf), int(totalPhotons), ptr(out)))
        return out

    def resetMoments(self):
        self._check(self._lib.mcbrat_reset_moments(self._ctx))

    def moments(self):
        buf = np.zeros(8 + 2 * self.momentsLength(), np.float64)
        self._check(self._lib.mcbrat_get_moments(self._ctx, ptr(buf)))
        return buf

    # -- measurement / parity ----------------------------------------------------------
    def enableCounters(self, on=True):
        self._check(self._lib.mcbrat_enable_counters(self._ctx, int(bool(on))))

    def counters(self):
        c = Counters()
        self._check(self._lib.mcbrat_get_counters(self._ctx, C.addressof(c)))
        d = c.as_dict()
        if getattr(self._lib, "_mcbrat_abi", 2) < 2:
            d["badPhotons"] = None  # (an A/B library from before ABI 2 does not write the field: unknown, not zero)
        return d

    def lastTraceMs(self):
        return float(self._lib.mcbrat_last_trace_ms(self._ctx))

    def traceFates(self, thisDomain, randomNumbers, incomingPhotons, n):
        self.specifyParameters()
        self._load_domain(thisDomain)
        self._load_source(incomingPhotons)
        fates = np.zeros(int(n), FATE_DTYPE)
        self._check(self._lib.mcbrat_trace_fates(self._ctx, randomNumbers.seed, randomNumbers.nextPhotonId, int(n), ptr(fates)))
        return fates


def new_Integrator(atmosphere, device=0):
    return Integrator(atmosphere, device)
