"""Phase functions and their tables: the inputs of the inverse-table builder.

Mirrors the parts of src/scatteringPhaseFunctions.f95 the path needs
(new_PhaseFunction :104-227, new_PhaseFunctionTable :229-440, getInfo_* :766-874)."""
import numpy as np

from ._capi import McbratError, lib, ptr


class PhaseFunction:
    """Either Legendre coefficients chi_1..chi_n (P0 = 1 implied) or (angle, value) pairs."""

    def __init__(self, legendreCoefficients=None, scatteringAngle=None, value=None,
                 extinction=1.0, singleScatteringAlbedo=1.0, description=""):
        if (legendreCoefficients is None) == (scatteringAngle is None):
            raise McbratError("newPhaseFunction: give Legendre coefficients or angle/value pairs")
        self.extinction, self.singleScatteringAlbedo, self.description = extinction, singleScatteringAlbedo, description
        if legendreCoefficients is not None:
            c = np.ascontiguousarray(legendreCoefficients, np.float32).reshape(-1)
            if c.size > 1 and (c[0] > 1.0 or c[0] < -1.0):  # :184-185
                raise McbratError("newPhaseFunction: Asymmetery parameter out of bounds.")
            self.legendreCoefficients, self.scatteringAngle, self.value = c, None, None
        else:
            a = np.ascontiguousarray(scatteringAngle, np.float32).reshape(-1)
            v = np.ascontiguousarray(value, np.float32).reshape(-1)
            if a.size != v.size:
                raise McbratError("newPhaseFunction: Number of scattering angles and phase function values must match.")
            if np.any(a < 0) or np.any(a > np.float32(np.pi) + np.spacing(np.float32(np.pi))):
                raise McbratError("newPhaseFunction: ScatteringAngle out of bounds.")
            if np.any(np.diff(a) <= 0):
                raise McbratError("newPhaseFunction: Scattering angle must be increasing, unique.")
            if np.any(v < 0):
                raise McbratError("newPhaseFunction: Negative phase function values supplied.")
            self.legendreCoefficients, self.scatteringAngle, self.value = None, a, v

    def inverse_table(self, nSteps):
        """computeInversePhaseFunction (src/inversePhaseFunctions.f95:66-174)."""
        out = np.zeros(nSteps, np.float32)
        if self.legendreCoefficients is not None:
            rc = lib().mcbrat_inverse_table_legendre(len(self.legendreCoefficients), ptr(self.legendreCoefficients),
                                                     nSteps, ptr(out))
        else:
            rc = lib().mcbrat_inverse_table_tabulated(len(self.scatteringAngle), ptr(self.scatteringAngle),
                                                      ptr(self.value), nSteps, ptr(out))
        if rc != 0:
            raise McbratError("computeInversePhaseFunctionTable: Can't compute inverse tables.")
        return out


    def forward_table(self, nAngles):
        """getPhaseFunctionValues at nAngles angles equally spaced on [0, pi] (tabulateForwardPhaseFunctions,
        src/opticalProperties.f95:1914-1916)."""
        out = np.zeros(nAngles, np.float32)
        if self.legendreCoefficients is not None:
            rc = lib().mcbrat_forward_table_legendre(len(self.legendreCoefficients), ptr(self.legendreCoefficients),
                                                     nAngles, ptr(out))
        else:
            rc = lib().mcbrat_forward_table_tabulated(len(self.scatteringAngle), ptr(self.scatteringAngle),
                                                      ptr(self.value), nAngles, ptr(out))
        if rc != 0:
            raise McbratError("tabulatePhaseFunctions: can't compute forward tables.")
        return out


class PhaseFunctionTable:
    def __init__(self, phaseFunctions, key=None, tableDescription=""):
        self.phaseFunctions = list(phaseFunctions)
        if not self.phaseFunctions:
            raise McbratError("newPhaseFunctionTable: no phase functions supplied")
        self.key = np.arange(1, len(self.phaseFunctions) + 1, dtype=np.float32) if key is None else \
            np.ascontiguousarray(key, np.float32)
        if len(self.key) != len(self.phaseFunctions):
            raise McbratError("newPhaseFunctionTable: Number of phase functions and key values must match.")
        if np.any(np.diff(self.key) <= 0):
            raise McbratError("newPhaseFunctionTable: Key values must be unique, increasing.")
        self.description = tableDescription

    @property
    def nEntries(self):
        return len(self.phaseFunctions)

    def inverse_table(self, nSteps):
        """computeInversePhaseFuncTable (:26-64): [nEntries, nSteps]."""
        return np.stack([p.inverse_table(nSteps) for p in self.phaseFunctions])


    def forward_table(self, nAngles):
        """[nEntries, nAngles]"""
        return np.stack([p.forward_table(nAngles) for p in self.phaseFunctions])


def computeHybridPhaseFunctions(values, GaussianWidth):
    """src/opticalProperties.f95:1937-2009 on values[nEntries, nAngles] (angles equally spaced on [0, pi])."""
    v = np.ascontiguousarray(values, np.float32)
    out = np.zeros_like(v)
    if lib().mcbrat_hybrid_phase_functions(v.shape[1], v.shape[0], ptr(v), float(GaussianWidth), ptr(out)) != 0:
        raise McbratError("computeHybridPhaseFunctions: invalid table or width")
    return out


def new_PhaseFunction(*args, **kw):
    if len(args) == 1:
        return PhaseFunction(legendreCoefficients=args[0], **kw)
    if len(args) == 2:
        return PhaseFunction(scatteringAngle=args[0], value=args[1], **kw)
    return PhaseFunction(**kw)


def new_PhaseFunctionTable(phaseFunctions, key=None, tableDescription=""):
    return PhaseFunctionTable(phaseFunctions, key, tableDescription)
