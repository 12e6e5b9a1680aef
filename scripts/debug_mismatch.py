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
    same = (got["fate"] == ref["fate"]) & (got["ix"] == ref["ix"]) & (got["iy"] == ref["iy"]) & (got["nScatter"] == ref["nScatter"]) & (np.abs(got["weight"] - ref["weight"]) <= 1e-6)
    bad = np.where(~same)[0]
    print(label, "identical %.4f" % same.mean(), "n bad", len(bad))
    print("  ref fate hist of bad:", np.bincount(ref["fate"][bad], minlength=4), " all:", np.bincount(ref["fate"], minlength=4))
    print("  ref nScatter of bad: mean %.1f  all: %.1f" % (ref["nScatter"][bad].mean() if len(bad) else 0, ref["nScatter"].mean()))
    first = np.array([min(g, r) for g, r in zip(got["nScatter"][bad], ref["nScatter"][bad])])
    print("  examples:", [(int(i), tuple(int(got[q][i]) for q in ("fate","ix","iy","nScatter")), tuple(int(ref[q][i]) for q in ("fate","ix","iy","nScatter")), float(got["weight"][i]), float(ref["weight"][i])) for i in bad[:5]])
    integ.finalize()

base = cases.stretched_grid_cloud()
run(base, 0.6, 75.0, 20000, "full")
c = cases.stretched_grid_cloud(); c["albedo"] = 0.0
run(c, 0.6, 75.0, 20000, "albedo0")
c = cases.stretched_grid_cloud(); c["albedo"] = 0.0
c["components"][0].pop("tabulated"); c["components"][0]["legendre"] = [cases.hg_legendre(0.8, 32), cases.hg_legendre(0.7, 32)]
run(c, 0.6, 75.0, 20000, "albedo0+legendre")
c = cases.stretched_grid_cloud(); c["albedo"] = 0.0
c["xe"] = np.linspace(0, c["xe"][-1], len(c["xe"])); c["ye"] = np.linspace(0, c["ye"][-1], len(c["ye"]))
run(c, 0.6, 75.0, 20000, "albedo0+uniform xy")
c = cases.stretched_grid_cloud(); c["albedo"] = 0.0
run(c, 1.0, 0.0, 20000, "albedo0+overhead sun")
