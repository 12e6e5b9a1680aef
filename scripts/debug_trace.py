import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from tests import cases
import mcbrat3d_amd as M
from mcbrat3d_amd.integrator import new_RandomNumberSequence
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from debug_mismatch3 import base  # noqa
case = base(ssa=0.6)
idx = int(sys.argv[1]); n = idx + 1
os.environ["MCBRAT_TRACE_PHOTON"] = str(idx); os.environ["ORC_TRACE_PHOTON"] = str(idx)
dom = cases.product_domain(case)
integ = M.new_Integrator(dom); integ.specifyParameters(minInverseTableSize=10001)
photons = M.new_PhotonStream(0.6, 75.0, numberOfPhotons=10 ** 9)
got = integ.traceFates(dom, new_RandomNumberSequence(20240917), photons, n)
P = cases.oracle_problem(case)
ref = O.compute_rt(P, O.solar_source(0.6, 75.0), O.philox_rng(20240917, 0), n, want_fates=True)["fates"]
print(got[idx], ref[idx])
