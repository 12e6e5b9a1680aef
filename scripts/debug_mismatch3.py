import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from tests import cases
import mcbrat3d_amd as M
from mcbrat3d_amd.integrator import new_RandomNumberSequence

def run(case, mu0, phi0, n, label):
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=10001)
    photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 9)
    got = integ.traceFates(dom, new_RandomNumberSequence(20240917), photons, n)
    P = cases.oracle_problem(case)
    ref = O.compute_rt(P, O.solar_source(mu0, phi0), O.philox_rng(20240917, 0), n, want_fates=True)["fates"]
    same = (got["fate"] == ref["fate"]) & (got["ix"] == ref["ix"]) & (got["iy"] == ref["iy"]) & (got["iz"] == ref["iz"]) & (got["nScatter"] == ref["nScatter"])
    bad = np.where(~same)[0]
    print(label, "identical %.5f" % same.mean(), "bad", len(bad), " mean nScatter %.2f" % ref["nScatter"].mean())
    for i in bad[:8]:
        print("    ", int(i), tuple(int(got[q][i]) for q in ("fate","ix","iy","iz","nScatter")), tuple(int(ref[q][i]) for q in ("fate","ix","iy","iz","nScatter")))
    integ.finalize()

def base(nx=96, ny=40, nz=18, ssa=0.0, seed=3):
    rng = np.random.default_rng(seed)
    xe = np.linspace(0, 0.8, nx + 1); ye = np.linspace(0, 0.4, ny + 1)
    ze = np.concatenate([[0.0], np.cumsum(0.03 * 1.1 ** np.arange(nz))])
    ext = np.exp(rng.normal(np.log(4.0), 0.9, (nx, ny, nz)))
    return dict(name="dbg", xe=xe, ye=ye, ze=ze, albedo=0.0,
                components=[dict(ext=ext, ssa=np.full_like(ext, ssa), pfIndex=np.ones(ext.shape, np.int32), legendre=[cases.hg_legendre(0.8, 32)])])

if __name__ == "__main__":
  n = 50000
  run(base(), 0.6, 75.0, n, "ssa=0 (one leg), slant sun")
  run(base(), 1.0, 0.0, n, "ssa=0 (one leg), overhead sun")
  run(base(ssa=0.6), 0.6, 75.0, n, "ssa=0.6 (few legs)")
