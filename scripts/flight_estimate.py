"""Analysis behind the clear-air flight, on the CPU: an instrumented COPY of the oracle's voxel walk
(built into ab/, git-ignored) classifies every cell a photon of the 128x128x64 workloads visits -- inside the cloud range
of its 4 x 4 brick column, outside it, or in a one-extinction layer -- and replays the kernel's rule (take off in a cell
outside the range, step per brick column, land when a range is entered) to count face-by-face steps, flight steps, and
flights per photon.  `erode` > 0 lets flights start only where the neighbouring brick columns are outside their ranges too.

usage: python scripts/flight_estimate.py landsat|radar [erode] [photons]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from tests import cases  # noqa: E402
from oracle import oracle as O  # noqa: E402


def build_instrumented():
    src = open(os.path.join(ROOT, "oracle", "mcbrat_oracle.c")).read()
    head = "float orc_accumulate_extinction(const orc_problem *P,"
    src = src.replace(head, """const signed char *g_cls = 0; long long g_stat[16];
void orc_set_cls(const signed char *c) { g_cls = c; for (int i = 0; i < 16; i++) g_stat[i] = 0; }
long long orc_get_stat(int i) { return g_stat[i]; }
""" + head, 1)
    loop = "  double z0 = P->ze[0], zMax = P->ze[P->nz];\n  for (;;) {\n    double step[3];"
    assert loop in src
    src = src.replace(loop, """  double z0 = P->ze[0], zMax = P->ze[P->nz];
  int flying = 0, prevBx = -1, prevBy = -1, first = 1;
  for (;;) {
    if (g_cls) { /* 0 inside the brick column's range, 1 outside (a flight may start), 2 one-extinction layer, 3 outside, no start */
      int c = g_cls[IDX3(P, idx[0], idx[1], idx[2])];
      int bx = (idx[0] - 1) >> 2, by = (idx[1] - 1) >> 2;
      if (flying) {
        if (c == 0) { flying = 0; g_stat[2]++; g_stat[0]++; }
        else if (bx != prevBx || by != prevBy) g_stat[1]++;
      } else {
        if (c == 1 || (c == 2 && first)) { flying = 1; g_stat[3]++; g_stat[1]++; }
        else g_stat[0]++;
      }
      first = 0; prevBx = bx; prevBy = by;
    }
    double step[3];""", 1)
    os.makedirs(os.path.join(ROOT, "ab"), exist_ok=True)
    cpath, so = os.path.join(ROOT, "ab", "orc_inst.c"), os.path.join(ROOT, "ab", "liborc_inst.so")
    open(cpath, "w").write(src)
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-std=c11", "-I" + os.path.join(ROOT, "oracle"),
                           "-shared", "-o", so, cpath, "-lm"])
    return so


def main():
    case_name = sys.argv[1] if len(sys.argv) > 1 else "landsat"
    erode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    nph = int(float(sys.argv[3])) if len(sys.argv) > 3 else 20000
    case = cases.landsat_like() if case_name == "landsat" else cases.radar_like()
    P = cases.oracle_problem(case)
    L = C.CDLL(build_instrumented())
    L.orc_compute_rt.restype = C.c_int64
    L.orc_compute_rt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 6
    L.orc_get_stat.restype = C.c_longlong
    nx, ny, nz = P.nx, P.ny, P.nz
    ext = np.asarray(P.totalExt).reshape(nz, ny, nx)
    bg, uniform = np.zeros(nz), np.zeros(nz, bool)
    for k in range(nz):
        v, c = np.unique(ext[k], return_counts=True)
        bg[k], uniform[k] = v[np.argmax(c)], len(v) == 1
    differs = ext != bg[:, None, None]
    nby, nbx = ny // 4, nx // 4
    lo, hi = np.full((nby, nbx), nz, int), np.zeros((nby, nbx), int)
    for by in range(nby):
        for bx in range(nbx):
            ks = np.nonzero(differs[:, by * 4:by * 4 + 4, bx * 4:bx * 4 + 4].any(axis=(1, 2)))[0]
            if len(ks):
                lo[by, bx], hi[by, bx] = ks.min(), ks.max() + 1
    cls = np.zeros((nz, ny, nx), np.int8)
    for k in range(nz):
        if uniform[k]:
            cls[k] = 2
            continue
        outside = (k < lo) | (k >= hi)
        far = outside.copy()
        for _ in range(erode):
            f2 = far.copy()
            for sy, sx in ((0, 1), (0, -1), (1, 0), (-1, 0), (1, 1), (1, -1), (-1, 1), (-1, -1)):
                f2 &= np.roll(np.roll(far, sy, 0), sx, 1)
            far = f2
        cls[k] = np.kron(np.where(far, 1, np.where(outside, 3, 0)), np.ones((4, 4), np.int8))
    flat = np.ascontiguousarray(cls.ravel())
    L.orc_set_cls(flat.ctypes.data_as(C.c_void_p))
    src, rng = O.solar_source(0.5, 30.0), O.philox_rng(7)
    ncol = nx * ny
    up, dn, ab = (np.zeros(ncol, np.float32) for _ in range(3))
    vol = np.zeros(ncol * nz, np.float32)
    cnt = O.OrcCounters()
    L.orc_compute_rt(C.addressof(P.c), C.addressof(src.c if hasattr(src, "c") else src), C.addressof(rng), nph,
                     up.ctypes.data, dn.ctypes.data, ab.ctypes.data, vol.ctypes.data, C.addressof(cnt), None)
    st = [L.orc_get_stat(i) / nph for i in range(4)]
    print("%s erode=%d: cells by class (in range / outside / one-extinction layer): %.3f %.3f %.3f" % (
        case_name, erode, (cls == 0).mean(), ((cls == 1) | (cls == 3)).mean(), (cls == 2).mean()))
    print("per photon: reference crossings %.1f; with the flight: face-by-face steps %.1f, flight steps %.1f, flights started %.2f, ended inside the domain %.2f" % (
        cnt.as_dict()["crossings"] / nph, st[0], st[1], st[3], st[2]))


if __name__ == "__main__":
    main()
