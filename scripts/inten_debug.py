import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import cases
from tests.test_gpu_intensity import _gpu, _oracle, _as_xyd
import mcbrat3d_amd as M
for name, kw in (("base", {}), ("ssa0", dict(ssa=0.0)), ("thin", dict(ext=0.05)), ("alb0", dict(albedo=0.0)), ("ssa0alb0", dict(ssa=0.0, albedo=0.0))):
    case = cases.homog_lw(n=6, **kw)
    mus, phis = [1.0, 0.5], [0.0, 0.0]
    n = 3000
    res, _, _ = _gpu(M, case, 1.0, 0.0, n, mus, phis, lw=True)
    ref = _oracle(case, 1.0, 0.0, n, mus, phis, lw=True)
    g, r = res["intensity"], _as_xyd(ref, 6, 6)
    print(name, res["meanIntensity"], ref["meanIntensity"], "max pix diff", np.max(np.abs(g - r)), "flux", res["meanFluxUp"], ref["meanFluxUp"], flush=True)
