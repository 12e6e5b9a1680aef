"""The oracle has two generator modes.  Its MT mode is the reference-faithful one: MT19937, the reference's
draw order, the rejection loop of next_direct (:1929-1936), u/(2^32-1) uniforms -- pinned to the reference's own
recorded outputs in tests/test_oracle_pin.py.  Its Philox mode is what the GPU parity tests compare the kernels
with photon by photon: it shares the kernel's stated deviations (one uniform for the azimuth, float(u) 2^-32,
fixed slot roles).  This test is the link between the two that the GPU tests lean on: fluxes, column fluxes and the
per-level absorption profile of the two modes agree to the parity statistic of SURVEY.md section 8d (CPU only)."""
import pytest

from tests import stats

PPB = 100000  # the reference tallies in float32: batches stay small


@pytest.mark.parametrize("name,make,kw,mu0,phi0,nb", [
    ("i3rcStepCloud", "step_cloud", dict(ssa=0.99), 1.0, 0.0, 24),
    ("i3rcStepCloud mu0=0.5", "step_cloud", dict(ssa=0.99), 0.5, 0.0, 20),
    ("cloud field 24x24x32, two components", "landsat_like", dict(n=24, nz=32), 0.5, 30.0, 20),
])
def test_philox_mode_agrees_with_mt_mode(name, make, kw, mu0, phi0, nb):
    a = stats.oracle_run(make, kw, "philox", nb, PPB, mu0, phi0, seed=10)
    b = stats.oracle_run(make, kw, "mt", nb, PPB, mu0, phi0, seed=10)
    A = {q: stats.mean_err(a[q]) for q in stats.QUANTITIES}
    B = {q: stats.mean_err(b[q]) for q in stats.QUANTITIES}
    rep = stats.assert_parity(A, B, name)
    print(rep)
