"""Photon sources.  Mirrors src/monteCarloIllumination.f95 (new_PhotonStream,
Directional :62-101 and BBEmission :431-522 forms) and the part of
src/emissionAndBroadBandWeights.f95 the thermal source needs (new_Weights,
emission_weighting :424-550).

The reference pre-generates every photon of a batch on the host (32 B/photon);
here a PhotonStream only carries the source description and a count -- photons are
generated on the GPU from the counter-based generator."""
import ctypes as C

import numpy as np

from ._capi import McbratError, lib, ptr


class Weights:
    """type(Weights): running voxel CDF + fraction of power emitted by the atmosphere."""

    def __init__(self, numX, numY, numZ, numLambda=1):
        self.numX, self.numY, self.numZ, self.numLambda = numX, numY, numZ, numLambda
        self._version = 0  # bumped by emission_weighting: integrators re-upload the voxel CDF when it moves
        self.voxelWeights = None
        self.fracAtmsPower = 0.0
        self.spectrIntgrFlux = 0.0


def new_Weights(numX, numY, numZ, numLambda=1):
    return Weights(numX, numY, numZ, numLambda)


def emission_weighting(thisDomain, theseWeights, sfcTemp, dLambda=1.0):
    """emission_weightingNEW: fills theseWeights, returns the emitted flux [W m-2]."""
    info = thisDomain.getInfo_Domain()
    if info["temps"] is None:
        raise McbratError("emission_weighting: domain has no temperatures")
    nx, ny, nz, nc = info["numX"], info["numY"], info["numZ"], info["numberOfComponents"]
    temps = np.ascontiguousarray(info["temps"].transpose(2, 1, 0)).reshape(-1)  # x fastest
    vw = np.zeros(nx * ny * nz, np.float64)
    frac, flux = C.c_double(), C.c_double()
    tot = np.ascontiguousarray(info["totalExt"]).reshape(-1)
    cum = np.ascontiguousarray(info["cumExt"]).reshape(-1)
    ssa = np.ascontiguousarray(info["ssa"]).reshape(-1)
    rc = lib().mcbrat_emission_weighting(nx, ny, nz, nc, ptr(info["xPosition"]), ptr(info["yPosition"]),
                                         ptr(info["zPosition"]), ptr(temps), ptr(tot), ptr(cum), ptr(ssa),
                                         info["albedo"], thisDomain.lambda_um, float(sfcTemp), float(dLambda),
                                         ptr(vw), C.byref(frac), C.byref(flux))
    if rc != 0:
        raise McbratError("emission_weightingNEW: Neither surface nor atmosphere will emitt photons since "
                          "total power is 0. Not a valid solution")
    theseWeights.voxelWeights, theseWeights.fracAtmsPower, theseWeights.spectrIntgrFlux = vw, frac.value, flux.value
    theseWeights._version += 1
    return flux.value


class PhotonStream:
    def __init__(self, numberOfPhotons, solarMu=None, solarAzimuth=None, theseWeights=None):
        if numberOfPhotons < 0:
            raise McbratError("setIllumination: must ask for non-negative number of photons.")
        self.numberOfPhotons = int(numberOfPhotons)
        self.currentPhoton = 1
        if theseWeights is not None:
            if theseWeights.voxelWeights is None:
                raise McbratError("setIllumination: weights have not been computed (call emission_weighting).")
            self.kind, self.weights = "BBEmission", theseWeights
        else:
            if solarAzimuth < 0.0 or solarAzimuth > 360.0:
                raise McbratError("setIllumination: solarAzimuth out of bounds")
            if abs(solarMu) > 1.0 or abs(solarMu) <= np.finfo(np.float32).tiny:
                raise McbratError("setIllumination: solarMu out of bounds")
            self.kind, self.solarMu, self.solarAzimuth = "Directional", float(solarMu), float(solarAzimuth)

    def morePhotonsExist(self):
        return 0 < self.currentPhoton <= self.numberOfPhotons


def new_PhotonStream(solarMu=None, solarAzimuth=None, numberOfPhotons=0, theseWeights=None):
    return PhotonStream(numberOfPhotons, solarMu, solarAzimuth, theseWeights)
