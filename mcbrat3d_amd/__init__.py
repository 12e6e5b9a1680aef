"""mcbrat3d_amd -- MI355X-native photon-tracing integrator (the hot path of MCBRaT3D).

Host-side mirror of the reference's interface for this path, over the C ABI in
include/mcbrat.h and hand-written HIP kernels for gfx950:

    scatteringPhaseFunctions -> mcbrat3d_amd.phase       (PhaseFunction, PhaseFunctionTable)
    opticalProperties        -> mcbrat3d_amd.domain      (Domain, addOpticalComponent, ...)
    monteCarloIllumination   -> mcbrat3d_amd.illumination (PhotonStream)
    emissionAndBBWeights     -> mcbrat3d_amd.illumination (Weights, emission_weighting)
    surfaceProperties        -> mcbrat3d_amd.surface     (SurfaceDescription)
    monteCarloRadiativeTransfer -> mcbrat3d_amd.integrator (Integrator)
    monteCarloDriver worker loop + statistics -> mcbrat3d_amd.driver
"""
from ._capi import McbratError  # noqa: F401
from .phase import PhaseFunction, PhaseFunctionTable, new_PhaseFunction, new_PhaseFunctionTable  # noqa: F401
from .domain import Domain, new_Domain  # noqa: F401
from .illumination import PhotonStream, Weights, new_PhotonStream, new_Weights, emission_weighting  # noqa: F401
from .integrator import Integrator, new_Integrator  # noqa: F401
from .surface import SurfaceDescription, new_SurfaceDescription  # noqa: F401
