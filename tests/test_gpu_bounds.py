"""The kernels cannot hang.  The reference's walk ends because it marches by cell index and
drops a photon whose step is not positive (opticalProperties.f95:1719-1722, counted in nBad,
monteCarloRadiativeTransfer.f95:562-563).  The kernels here find cells from positions in places and keep face
distances in float, so every loop carries a bound of its own; a photon that exceeds one is dropped and counted in
counters()["badPhotons"].  These tests drive the kernels into their bounds on purpose: the three soak finds of round 2
with the ORIGINAL (pre-fix) tie handling switched back on (MCBRAT_TEST_LEGACY_TIES, a test-only switch), a medium
whose inverse table holds a NaN under conservative scattering (which the reference itself never finishes), and a leg
budget set low.  Run on the MI355X box with `-m gpu`."""
import numpy as np
import pytest

from tests import cases
from tests.test_gpu_block_walk import random_box_case
from tests.test_gpu_parity import random_oracle_case

pytestmark = pytest.mark.gpu
SEED = 90210


@pytest.fixture(scope="module")
def M():
    import mcbrat3d_amd
    return mcbrat3d_amd


def _run(M, case, mu0, phi0, rr, n, tables=9001, **tuning):
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    dom = cases.product_domain(case)
    integ = M.new_Integrator(dom)
    integ.specifyParameters(minInverseTableSize=tables, useRayTracing=True, useRussianRoulette=rr)
    integ.setTuning(eventThreshold=16, **tuning)
    photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 9)
    fates = integ.traceFates(dom, new_RandomNumberSequence(SEED), photons, n)
    bad_fates = int(integ.counters()["badPhotons"])
    integ.resetMoments()
    integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, n)
    res = integ.reportResults()
    bad = int(integ.counters()["badPhotons"])
    integ.finalize()
    return fates, res, bad_fates, bad - bad_fates


# (test, soak seed, bit of MCBRAT_TEST_LEGACY_TIES): what each fix of round 2 replaced
LEGACY = [("box", 168, 1),      # cell look-ups of a block crossing not clamped to the block the lane is leaving
          ("oracle", 71, 2),    # a photon with a NaN direction moved along it
          ("oracle", 763, 4)]   # "spans a periodic axis" bits kept after the fold


@pytest.mark.allow_bad_photons
@pytest.mark.timeout(120, method="thread")
@pytest.mark.parametrize("which,seed,ties", LEGACY)
def test_soak_finds_end_with_the_original_tie_handling(M, which, seed, ties, monkeypatch):
    """Seeds 168, 71 and 763 hung the block walk before their fixes.  With the pre-fix handling switched back on the
    kernel must still END: the stuck photon is dropped by the bound on a leg's block crossings (or the leg budget) and counted, everybody
    else's history is what the fixed kernel gives."""
    case, mu0, phi0, rr = random_box_case(seed) if which == "box" else random_oracle_case(seed)
    n = 20000 if which == "box" else 15000
    tables = 2001 if which == "box" else 9001
    good = _run(M, case, mu0, phi0, rr, n, tables, blockWalk=2)
    assert good[2] == 0 and good[3] == 0
    monkeypatch.setenv("MCBRAT_TEST_LEGACY_TIES", str(ties))  # read when the context is created
    monkeypatch.setenv("MCBRAT_WATCHDOG", "50000")             # (the default, 2^20 block crossings of one leg, takes about a second to reach)
    monkeypatch.setenv("MCBRAT_MAX_EVENTS", "200000")          # (the default, 2^24 legs, takes many seconds for a lone photon to reach)
    old = _run(M, case, mu0, phi0, rr, n, tables, blockWalk=2)
    dropped = old[0]["fate"] == 3
    # (the switch is compiled into the instrumented instantiation only -- the one traceFates runs; the production kernel
    # of the second call ignores it and drops nothing)
    assert dropped.sum() == old[2] and old[3] == 0, (dropped.sum(), old[2], old[3])
    assert old[2] <= 8, old[2]
    keep = ~dropped
    same = (old[0]["fate"][keep] == good[0]["fate"][keep]) & (old[0]["nScatter"][keep] == good[0]["nScatter"][keep])
    assert same.mean() > 0.99, same.mean()
    print("legacy ties %d, seed %d: %d photon(s) dropped by the bounds" % (ties, seed, old[2]))


@pytest.mark.allow_bad_photons
@pytest.mark.timeout(120, method="thread")
@pytest.mark.parametrize("walk,tuning", [("face by face", dict(privateTallies=0, layerSkip=0)), ("LDS face by face", dict(blockWalk=0)),
                                         ("block walk", dict(blockWalk=2))])
def test_nan_table_entry_under_conservative_scattering_ends(M, walk, tuning, monkeypatch):
    """computeInversePhaseFunction can leave a NaN in the table (one entry in 9001 for a 64-term HG
    series with g = 0.5).  A photon that draws it has a NaN direction and collides on in its cell; with omega0 = 1 its
    weight never falls, so neither roulette nor the reference's loop would ever end it.  Here it is dropped after
    2^20 legs -- about a second per photon, the production kernels' constant; 4096 in the instrumented instantiation of this
    test (MCBRAT_MAX_EVENTS_NAN) -- and counted; the energy it carried is the only energy missing."""
    from oracle import oracle as O
    monkeypatch.setenv("MCBRAT_MAX_EVENTS_NAN", "4096")
    leg = cases.hg_legendre(0.5, 64)
    assert np.isnan(O.inverse_table_legendre(leg, 9001)).any()  # (the premise: this table does hold a NaN)
    nx = nz = 8
    ext = np.full((nx, 1, nz), 40.0)
    case = dict(name="nanTable", xe=np.linspace(0.0, 0.4, nx + 1), ye=np.array([0.0, 0.4]), ze=np.linspace(0.0, 0.4, nz + 1), albedo=0.0,
                components=[dict(ext=ext, ssa=np.ones(ext.shape), pfIndex=np.ones(ext.shape, np.int32), legendre=[leg])])
    n = 20000
    fates, res, bad_f, bad_c = _run(M, case, 1.0, 0.0, True, n, 9001, **tuning)
    assert bad_f >= 1 and bad_f == bad_c == int((fates["fate"] == 3).sum()), (walk, bad_f, bad_c, (fates["fate"] == 3).sum())
    closure = res["meanFluxUp"] + res["meanFluxDown"] + res["meanFluxAbsorbed"]
    assert abs(closure + bad_c / n - 1.0) < 1e-3, (walk, closure, bad_c / n)


@pytest.mark.allow_bad_photons
@pytest.mark.timeout(120, method="thread")
@pytest.mark.parametrize("block_walk", [0, 1])
def test_leg_budget(M, block_walk, monkeypatch):
    """The leg budget (2^24 legs per photon: a compile-time constant of the production kernels; the instrumented instantiation
    takes it from MCBRAT_MAX_EVENTS) set to 16 on the conservative step cloud: the photons that need more legs are dropped
    with fate 3 and counted; the others are untouched, and so is the production run of the same photons."""
    case = cases.step_cloud(ssa=1.0)
    n = 20000
    full = _run(M, case, 1.0, 0.0, True, n, 10001, blockWalk=block_walk)
    assert full[2] == 0 and full[3] == 0
    monkeypatch.setenv("MCBRAT_MAX_EVENTS", "16")
    cut = _run(M, case, 1.0, 0.0, True, n, 10001, blockWalk=block_walk)
    long_ones = full[0]["nEvents"] > 16
    assert long_ones.sum() > 100
    assert np.array_equal(cut[0]["fate"] == 3, long_ones)
    assert cut[2] == int(long_ones.sum()) and cut[3] == 0
    short = ~long_ones
    assert np.array_equal(cut[0]["fate"][short], full[0]["fate"][short]) and np.array_equal(cut[0]["nScatter"][short], full[0]["nScatter"][short])
