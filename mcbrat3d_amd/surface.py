"""Host-side mirror of src/surfaceProperties.f95: a Lambertian surface whose reflectance varies with horizontal
position (the reference's template for "a few parameters per patch" surface models has exactly one parameter)."""
import numpy as np

from ._capi import McbratError

numberOfParameters = 1  # :28


class SurfaceDescription:
    def __init__(self, xPosition, yPosition, BRDFParameters):
        self.xPosition = xPosition
        self.yPosition = yPosition
        self.BRDFParameters = BRDFParameters  # [numberOfParameters, numX - 1, numY - 1]

    def isReady_surfaceDescription(self):  # :165-172
        return self.xPosition is not None and self.yPosition is not None and self.BRDFParameters is not None


def new_SurfaceDescription(surfaceParameters, xPosition=None, yPosition=None):
    """newSurfaceDescriptionXY (:58-94) when positions are given, newSurfaceUniform (:96-115) otherwise."""
    params = np.asarray(surfaceParameters, np.float32)
    if xPosition is None and yPosition is None:
        if params.reshape(-1).size != numberOfParameters:
            raise McbratError("new_SurfaceDescription: Wrong number of parameters supplied for surface BRDF.")
        huge = float(np.finfo(np.float32).max)
        xPosition, yPosition = (0.0, huge), (0.0, huge)
        params = params.reshape(numberOfParameters, 1, 1)
    x = np.ascontiguousarray(xPosition, np.float64)
    y = np.ascontiguousarray(yPosition, np.float64)
    if params.ndim != 3 or params.shape[0] != numberOfParameters:
        raise McbratError("new_SurfaceDescription: Wrong number of parameters supplied for surface BRDF.")
    if params.shape[1] != x.size - 1 or params.shape[2] != y.size - 1:
        raise McbratError("new_SurfaceDescription: position vector(s) are incorrect length.")
    if np.any(np.diff(x) <= 0.0) or np.any(np.diff(y) <= 0.0):
        raise McbratError("new_SurfaceDescription: positions must be unique, increasing.")
    if np.any(params[0] < 0.0) or np.any(params[0] > 1.0):
        raise McbratError("new_SurfaceDescription: surface reflectance must be between 0 and 1")
    return SurfaceDescription(x, y, params.copy())
