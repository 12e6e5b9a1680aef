"""Flat binary domain file: what fortran/mcbrat_driver.f90 reads in place of the reference's
NetCDF .dom / SSP files (no NetCDF library exists in this image).  Layout (little endian,
Fortran stream order, arrays x fastest):

    int32 magic 'DMCM' (1296257860), nx, ny, nz, nc;  float64 surfaceAlbedo
    float64 xEdges[nx+1], yEdges[ny+1], zEdges[nz+1]
    float64 totalExt[nvox], cumExt[nc*nvox], ssa[nc*nvox];  int32 phaseFuncIndex[nc*nvox]
    per component: int32 nSteps, nEntries; float32 table[nEntries][nSteps]
"""
import numpy as np

MAGIC = 1296257860


def write_flat_domain(path, domain, tableSize=10001):
    info = domain.getInfo_Domain()
    tables = domain.tabulateInversePhaseFunctions(tableSize)
    with open(path, "wb") as f:
        np.array([MAGIC, info["numX"], info["numY"], info["numZ"], info["numberOfComponents"]], "<i4").tofile(f)
        np.array([info["albedo"]], "<f8").tofile(f)
        for k in ("xPosition", "yPosition", "zPosition", "totalExt", "cumExt", "ssa"):
            np.ascontiguousarray(info[k], "<f8").reshape(-1).tofile(f)
        np.ascontiguousarray(info["phaseFuncI"], "<i4").reshape(-1).tofile(f)
        for t in tables:
            t = np.ascontiguousarray(t, "<f4")
            np.array([t.shape[1], t.shape[0]], "<i4").tofile(f)
            t.reshape(-1).tofile(f)
    return path
