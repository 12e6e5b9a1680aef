"""Per-photon divergence study (development; results quoted in DESIGN.md section 3).

Same Philox streams in the HIP kernel and the CPU oracle give identical photon histories only while
float32 rounding of the accumulated optical depth cannot move a ray across a boundary between cells
of different extinction.  This script measures the identical fraction for media of increasing
heterogeneity, and (--trace N) prints the per-collision records of photon N from both sides.

    python scripts/chaos_study.py [--trace 6874]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import oracle as O  # noqa: E402
from tests import cases  # noqa: E402
import mcbrat3d_amd as M  # noqa: E402
from mcbrat3d_amd.integrator import new_RandomNumberSequence  # noqa: E402

SEED = 20240917


def medium(nx=24, ny=10, nz=18, extmode="cell", ssa=0.97, seed=3):
    rng = np.random.default_rng(seed)
    xe, ye = np.linspace(0, 0.8, nx + 1), np.linspace(0, 0.4, ny + 1)
    ze = np.concatenate([[0.0], np.cumsum(0.03 * 1.1 ** np.arange(nz))])
    if extmode == "cell":
        ext = np.exp(rng.normal(np.log(4.0), 0.9, (nx, ny, nz)))
    elif extmode == "column":
        ext = np.exp(rng.normal(np.log(4.0), 0.9, (nx, ny, 1))) * np.ones((1, 1, nz))
    else:
        ext = np.full((nx, ny, nz), 4.0)
    return dict(name="chaos", xe=xe, ye=ye, ze=ze, albedo=0.0,
                components=[dict(ext=ext, ssa=np.full_like(ext, ssa), pfIndex=np.ones(ext.shape, np.int32),
                                 legendre=[cases.hg_legendre(0.8, 32)])])


def compare(case, mu0, phi0, n, label):
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=10001)
    photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 9)
    got = integ.traceFates(dom, new_RandomNumberSequence(SEED), photons, n)
    ref = O.compute_rt(cases.oracle_problem(case), O.solar_source(mu0, phi0), O.philox_rng(SEED, 0), n, want_fates=True)["fates"]
    same = (got["fate"] == ref["fate"]) & (got["ix"] == ref["ix"]) & (got["iy"] == ref["iy"]) & (got["iz"] == ref["iz"]) & \
        (got["nScatter"] == ref["nScatter"])
    print("%-44s identical %.5f   mean scattering order %.1f" % (label, same.mean(), ref["nScatter"].mean()))
    integ.finalize()
    return got, ref


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", type=int, default=-1)
    a = ap.parse_args()
    if a.trace >= 0:
        os.environ["MCBRAT_TRACE_PHOTON"] = os.environ["ORC_TRACE_PHOTON"] = str(a.trace)
        g, r = compare(medium(96, 40, ssa=0.6), 0.6, 75.0, a.trace + 1, "traced photon (records on stderr)")
        print(g[a.trace], r[a.trace])
    else:
        n = 20000
        compare(medium(extmode="const"), 1.0, 0.0, n, "constant extinction")
        compare(medium(extmode="column"), 1.0, 0.0, n, "extinction varies by column")
        compare(medium(), 1.0, 0.0, n, "extinction varies by cell (33 m cells)")
        compare(medium(96, 40), 1.0, 0.0, n, "extinction varies by cell (8 m cells)")
        compare(medium(96, 40, ssa=0.0), 0.6, 75.0, 50000, "first leg only (ssa = 0), 8 m cells")
