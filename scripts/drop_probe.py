"""Which photon a loop bound dropped, and why: the random layering of tests/test_gpu_flight.py for one seed, per walk.
usage: python scripts/drop_probe.py <seed>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from tests import cases  # noqa: E402
import tests.test_gpu_flight as T  # noqa: E402
import mcbrat3d_amd as M  # noqa: E402

seed = int(sys.argv[1])
captured = {}
orig_run = T._run


def run(M_, case, mu0, phi0, skip, n, **kw):
    out = orig_run(M_, case, mu0, phi0, skip, n, **kw)
    captured[skip] = (case, mu0, phi0, out)
    return out


T._run = run
try:
    T.test_random_layerings_against_face_by_face_walk.__wrapped__(M, seed) if hasattr(T.test_random_layerings_against_face_by_face_walk, "__wrapped__") \
        else T.test_random_layerings_against_face_by_face_walk(M, seed)
except AssertionError as e:
    print("assertion:", str(e)[:200])
from oracle import oracle as O  # noqa: E402
for skip, (case, mu0, phi0, out) in captured.items():
    f = out["fates"]
    d = np.flatnonzero(f["fate"] == 3)
    print("skip", skip, "mu0", mu0, "phi0", phi0, "grid", len(case["xe"]) - 1, len(case["ye"]) - 1, len(case["ze"]) - 1, "dropped", d, f[d])
    leg = case["components"][0]["legendre"][0]
    print("   NaN in the inverse table:", int(np.isnan(O.inverse_table_legendre(leg, 9001)).sum()), "ssa", float(case["components"][0]["ssa"].max()),
          "layers (mean ext):", np.round(case["components"][0]["ext"].mean(axis=(0, 1)), 2))
