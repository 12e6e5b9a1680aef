"""The optical domain: grid + components expanded to full 3-D arrays.

Mirrors src/opticalProperties.f95: new_Domain (:455-552), addOpticalComponent
(:554-700), getOpticalPropertiesByComponent (:966-1072), getInfo_Domain (:796-962),
tabulateInversePhaseFunctions (:1817-1870).  Arrays handed to the integrator are
Fortran order (x fastest), the layout the C ABI takes."""
import numpy as np

from ._capi import McbratError


class Domain:
    def __init__(self, xPosition, yPosition, zPosition, temps=None, surfaceAlbedo=0.0, lambda_um=0.0):
        self.xPosition = np.ascontiguousarray(xPosition, np.float64)
        self.yPosition = np.ascontiguousarray(yPosition, np.float64)
        self.zPosition = np.ascontiguousarray(zPosition, np.float64)
        for n, e in (("x", self.xPosition), ("y", self.yPosition), ("z", self.zPosition)):
            if e.ndim != 1 or e.size < 2 or np.any(np.diff(e) <= 0):
                raise McbratError("new_Domain: %s positions must be increasing, unique." % n)  # :470-476
        if not (0.0 <= surfaceAlbedo <= 1.0):
            raise McbratError("new_Domain: surfaceAlbedo must be between 0 and 1.")
        self._version = 0  # bumped by whatever changes the optics: integrators re-upload when it moves
        self.surfaceAlbedo = float(surfaceAlbedo)
        self.lambda_um = float(lambda_um)
        self.numX, self.numY, self.numZ = len(self.xPosition) - 1, len(self.yPosition) - 1, len(self.zPosition) - 1
        self.temps = None if temps is None else np.ascontiguousarray(temps, np.float64).reshape(self.numX, self.numY, self.numZ)
        self.components = []
        self.totalExt = self.cumulativeExt = self.ssa = self.phaseFunctionIndex = None
        self.forwardTables = []
        self.inversePhaseFunctions = None

    @property
    def surfaceAlbedo(self):
        return self._surfaceAlbedo

    @surfaceAlbedo.setter
    def surfaceAlbedo(self, value):
        if not (0.0 <= value <= 1.0):
            raise McbratError("new_Domain: surfaceAlbedo must be between 0 and 1.")
        self._surfaceAlbedo = float(value)
        self._version += 1

    # addOpticalComponent3D / 1D (:554-700) with validateOpticalComponent (:1530-1591)
    def addOpticalComponent(self, componentName, extinction, singleScatteringAlbedo, phaseFunctionIndex,
                            phaseFunctions, zLevelBase=1):
        e = np.asarray(extinction, np.float64)
        s = np.asarray(singleScatteringAlbedo, np.float64)
        p = np.asarray(phaseFunctionIndex, np.int32)
        if e.shape != s.shape or e.shape != p.shape:
            raise McbratError("validateOpticalComponent: optical property grids must be the same size.")
        if e.ndim == 3:
            if e.shape[:2] != (self.numX, self.numY):
                raise McbratError("validateOpticalComponent: arrays don't span the horizontal extent of the domain.")
            nzc = e.shape[2]
        elif e.ndim == 1:
            nzc = e.shape[0]
        else:
            raise McbratError("validateOpticalComponent: extinction must be 1-D (z) or 3-D (x, y, z).")
        if zLevelBase < 1 or zLevelBase + nzc - 1 > self.numZ:
            raise McbratError("validateOpticalComponent: arrays don't fit in the vertical extent of the domain.")
        if np.any(e < 0):
            raise McbratError("validateOpticalComponent: extinction must be >= 0.")
        if np.any(s < 0) or np.any(s > 1):
            raise McbratError("validateOpticalComponent: singleScatteringAlbedo must be between 0 and 1.")
        if np.any(p[e > 0] < 1) or np.any(p > phaseFunctions.nEntries):
            raise McbratError("validateOpticalComponent: phaseFunctionIndex refers to entries that don't exist.")
        self.components.append(dict(name=componentName, ext=e, ssa=s, pfIndex=p, table=phaseFunctions,
                                    zLevelBase=int(zLevelBase)))
        self.totalExt = None  # expansion is stale
        self.inversePhaseFunctions = None
        self._version += 1

    def getOpticalPropertiesByComponent(self):
        """:966-1072: per-component fields expanded to (x, y, z, component)."""
        if not self.components:
            raise McbratError("getOpticalPropertiesByComponent: domain contains no optical components.")
        nc, nx, ny, nz = len(self.components), self.numX, self.numY, self.numZ
        cum = np.zeros((nc, nz, ny, nx), np.float64)
        ssa = np.zeros((nc, nz, ny, nx), np.float64)
        pfi = np.ones((nc, nz, ny, nx), np.int32)
        for i, comp in enumerate(self.components):
            lo = comp["zLevelBase"] - 1
            if comp["ext"].ndim == 1:  # horizontally uniform :1033-1043
                hi = lo + comp["ext"].shape[0]
                cum[i, lo:hi] = comp["ext"][:, None, None]
                ssa[i, lo:hi] = comp["ssa"][:, None, None]
                pfi[i, lo:hi] = comp["pfIndex"][:, None, None]
            else:
                hi = lo + comp["ext"].shape[2]
                cum[i, lo:hi] = comp["ext"].transpose(2, 1, 0)
                ssa[i, lo:hi] = comp["ssa"].transpose(2, 1, 0)
                pfi[i, lo:hi] = comp["pfIndex"].transpose(2, 1, 0)
        for i in range(1, nc):  # :1055-1057
            cum[i] += cum[i - 1]
        total = cum[nc - 1].copy()
        mask = total > np.finfo(np.float64).tiny  # :1059-1061
        for i in range(nc):
            cum[i][mask] /= total[mask]
        self.totalExt, self.cumulativeExt, self.ssa, self.phaseFunctionIndex = total, cum, ssa, pfi
        self.forwardTables = [c["table"] for c in self.components]
        return self

    def tabulateInversePhaseFunctions(self, tableSize):
        """:1817-1870: cached on the domain, rebuilt when a finer table is asked for."""
        if self.totalExt is None:
            self.getOpticalPropertiesByComponent()
        if self.inversePhaseFunctions is None or any(t.shape[1] < tableSize for t in self.inversePhaseFunctions):
            self.inversePhaseFunctions = [t.inverse_table(tableSize) for t in self.forwardTables]
        return self.inversePhaseFunctions

    def tabulateForwardPhaseFunctions(self, tableSize, hybrid=False, hybridWidth=0.0):
        """:1872-1935: (tabulatedPhaseFunctions, tabulatedOrigPhaseFunctions), each a list per component of
        [nEntries, tableSize] arrays; the first holds the hybrid versions when asked for."""
        from .phase import computeHybridPhaseFunctions
        key = (int(tableSize), bool(hybrid), float(hybridWidth))
        if getattr(self, "_fwd_key", None) != key:
            orig = [t.forward_table(int(tableSize)) for t in self.forwardTables]
            tab = [computeHybridPhaseFunctions(t, hybridWidth) for t in orig] if hybrid and hybridWidth > 0 else orig
            self._fwd_key, self._fwd = key, (tab, orig)
        return self._fwd

    def getInfo_Domain(self):
        if self.totalExt is None:
            self.getOpticalPropertiesByComponent()
        return dict(numX=self.numX, numY=self.numY, numZ=self.numZ, albedo=self.surfaceAlbedo,
                    numberOfComponents=len(self.components), xPosition=self.xPosition, yPosition=self.yPosition,
                    zPosition=self.zPosition, totalExt=self.totalExt, cumExt=self.cumulativeExt, ssa=self.ssa,
                    phaseFuncI=self.phaseFunctionIndex, temps=self.temps,
                    componentNames=[c["name"] for c in self.components])


def new_Domain(xPosition, yPosition, zPosition, temps=None, surfaceAlbedo=0.0, lambda_um=0.0):
    return Domain(xPosition, yPosition, zPosition, temps, surfaceAlbedo, lambda_um)
