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


@pytest.mark.allow_bad_photons
@pytest.mark.timeout(180, method="thread")
def test_wave_watchdog_of_the_face_by_face_kernel(M, monkeypatch):
    """trace_kernel's watchdog: a wave in which no lane has started a leg or taken a photon for
    `watchdog` consecutive event phases drops every photon it still holds -- each has then been on ONE leg for that many
    phases (up to 256 walk iterations each).  Vacuum over a Lambertian surface, 64 x 64 columns: the few photons that are
    reflected at a grazing angle cross thousands of cell faces on their way up.  With the watchdog at 16 phases they are
    dropped when nothing else is left in their wave; dropped photons come back with fate 3 and are counted, and everybody
    else's history is what the run with the default watchdog (2^20 phases: never reached) gives."""
    from mcbrat3d_amd.integrator import new_RandomNumberSequence
    nx, nz = 64, 4
    ext = np.zeros((nx, nx, nz))
    case = dict(name="vacuumOverSurface", xe=np.linspace(0.0, 1.0, nx + 1), ye=np.linspace(0.0, 1.0, nx + 1), ze=np.linspace(0.0, 1.0, nz + 1), albedo=0.5,
                components=[dict(ext=ext, ssa=np.zeros(ext.shape), pfIndex=np.ones(ext.shape, np.int32), legendre=[cases.hg_legendre(0.85, 16)])])
    n = 200000

    def run():
        dom = cases.product_domain(case)
        integ = M.new_Integrator(dom)
        integ.specifyParameters(minInverseTableSize=2001, useRayTracing=True, useRussianRoulette=False)
        integ.setTuning(eventThreshold=16, privateTallies=0, layerSkip=0, blockWalk=0)
        photons = M.new_PhotonStream(0.5, 0.0, numberOfPhotons=10 ** 9)
        fates = integ.traceFates(dom, new_RandomNumberSequence(SEED), photons, n)
        bad, what = int(integ.counters()["badPhotons"]), integ.firstDrop()
        integ.finalize()
        return fates, bad, what

    full, bad0, what0 = run()
    assert bad0 == 0 and what0 == "" and not (full["fate"] == 3).any()
    monkeypatch.setenv("MCBRAT_WATCHDOG", "16")  # (read when the context is created; the instrumented instantiation takes it from DevParams)
    cut, bad, what = run()
    assert "watchdog" in what and "trace_kernel" in what, what  # (the record of the first drop names the bound that fired)
    dropped = cut["fate"] == 3
    assert bad == int(dropped.sum()) and bad >= 1, (bad, int(dropped.sum()))
    assert bad < n // 100  # (only the grazing ones: a wave that still launches or scatters resets the count)
    for f in full.dtype.names:
        assert np.array_equal(cut[f][~dropped], full[f][~dropped], equal_nan=True), f
    # what was dropped were reflected photons on a long way up (in the full run they leave through the top after one reflection)
    assert np.all(full["fate"][dropped] == 0) and np.all(full["nScatter"][dropped] == 1)
    print("watchdog 16: %d of %d photons dropped" % (bad, n))


@pytest.mark.allow_bad_photons
@pytest.mark.timeout(120, method="thread")
def test_view_ray_length_bound(M, monkeypatch):
    """The radiance kernels end a view ray that is longer than any ray of the geometry can be (DevParams::rayMaxLen, the
    mitigation for a ray that circles the periodic domain) and count it.  Vacuum over a Lambertian
    surface: every photon is reflected once and sends one ray per view direction from the surface, of length H / mu.  With
    the bound forced (MCBRAT_TEST_RAY_MAX_LEN, test only) between the two views' lengths every ray of the slanted view is cut
    -- counted: exactly one per photon -- and the vertical view is untouched: A / pi exactly, as without the bound."""
    from tests.test_gpu_intensity import _gpu
    case = cases.plane_parallel(ssa=1.0)
    case["components"][0]["ext"] = np.zeros_like(case["components"][0]["ext"])
    case["albedo"] = 0.3
    height = float(case["ze"][-1] - case["ze"][0])
    n = 5000
    from mcbrat3d_amd import integrator as I
    seen = []
    orig = I.Integrator.finalize

    what = []

    def finalize(self):  # (the helper finalises its integrator: read the count first)
        if getattr(self, "_ctx", None):
            seen.append(int(self.counters()["badPhotons"]))
            what.append(self.firstDrop())
        orig(self)
    monkeypatch.setattr(I.Integrator, "finalize", finalize)
    for rr in (False, True):
        seen.clear()
        res, _, _ = _gpu(M, case, 0.7, 30.0, n, [1.0, 0.5], [0.0, 90.0], useRussianRouletteForIntensity=rr)
        assert seen[-1] == 0 and np.allclose(res["meanIntensity"], 0.3 / np.pi, rtol=1e-6)
        monkeypatch.setenv("MCBRAT_TEST_RAY_MAX_LEN", repr(1.5 * height))  # vertical rays: H; rays of the view at mu = 0.5: 2 H
        res, _, _ = _gpu(M, case, 0.7, 30.0, n, [1.0, 0.5], [0.0, 90.0], useRussianRouletteForIntensity=rr)
        monkeypatch.delenv("MCBRAT_TEST_RAY_MAX_LEN")
        assert seen[-1] == n, (rr, seen[-1])
        assert "view ray" in what[-1] and "view direction 1" in what[-1], what[-1]
        assert np.allclose(res["meanIntensity"][0], 0.3 / np.pi, rtol=1e-6) and res["meanIntensity"][1] == 0.0, (rr, res["meanIntensity"])


TEST_BOUNDS_LIB = "libmcbrat_testbounds.so"  # built by __graft_entry__.build() / here on demand: -DMCBRAT_TEST_BOUNDS_IN_PRODUCTION

_PRODUCTION_BOUNDS_SCRIPT = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.environ["MCBRAT_REPO"])
try:
    import torch  # noqa: F401  (one HIP runtime: conftest's order)
except Exception:
    pass
import mcbrat3d_amd as M
from mcbrat3d_amd.integrator import new_RandomNumberSequence
from tests import cases
from tests.test_gpu_block_walk import random_box_case
from tests.test_gpu_parity import random_oracle_case
SEED = 90210
which, seed, n, tables = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
if which == "box":
    case, mu0, phi0, rr = random_box_case(seed)
elif which == "oracle":
    case, mu0, phi0, rr = random_oracle_case(seed)
else:
    case, mu0, phi0, rr = cases.step_cloud(ssa=1.0), 1.0, 0.0, True
dom = cases.product_domain(case)
integ = M.new_Integrator(dom)
integ.specifyParameters(minInverseTableSize=tables, useRayTracing=True, useRussianRoulette=rr)
integ.setTuning(eventThreshold=16, blockWalk=int(sys.argv[5]))
photons = M.new_PhotonStream(mu0, phi0, numberOfPhotons=10 ** 9)
fates = integ.traceFates(dom, new_RandomNumberSequence(SEED), photons, n)          # the instrumented instantiation
bad_fates = int(integ.counters()["badPhotons"])
integ.resetMoments()
integ.computeRadiativeTransfer(dom, new_RandomNumberSequence(SEED), photons, n)   # the PRODUCTION instantiation, same photons
res = integ.reportResults()
bad = int(integ.counters()["badPhotons"]) - bad_fates
integ.finalize()
print(json.dumps(dict(dropped_fates=int((fates["fate"] == 3).sum()), bad_instrumented=bad_fates, bad_production=bad,
                      closure=float(res["meanFluxUp"] + res["meanFluxDown"] + res["meanFluxAbsorbed"]),
                      long_ones=int((fates["nEvents"] > 16).sum()) if which == "step" else None)))
"""


def _test_bounds_library():
    import os
    from mcbrat3d_amd import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "ab", TEST_BOUNDS_LIB)
    deps = [os.path.join(build.CSRC, d) for d in build.DEPS]
    if not os.path.exists(path) or any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        build.build(force=True, extra_flags=["-DMCBRAT_TEST_BOUNDS_IN_PRODUCTION"], out=path)  # (a few minutes: normally done by __graft_entry__.build())
    return root, path


def _run_with_test_bounds(env, *args):
    import json
    import os
    import subprocess
    import sys
    root, lib = _test_bounds_library()
    e = dict(os.environ, MCBRAT_LIB=lib, MCBRAT_REPO=root, **env)  # (MCBRAT_LIB is read when mcbrat3d_amd._capi is imported: a process of its own)
    out = subprocess.run([sys.executable, "-c", _PRODUCTION_BOUNDS_SCRIPT] + [str(a) for a in args], env=e, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.allow_bad_photons
@pytest.mark.timeout(900, method="thread")
@pytest.mark.parametrize("which,seed,ties", LEGACY)
def test_production_kernels_end_with_the_original_tie_handling(which, seed, ties):
    """The bounds of the PRODUCTION instantiations (compile-time constants in the shipped library; per-lane drop counter, a
    bound on a leg's block crossings instead of a watchdog) are other code than the instrumented instantiation's.  A test-only
    build in which they take the same test values (-DMCBRAT_TEST_BOUNDS_IN_PRODUCTION, ab/libmcbrat_testbounds.so) runs the
    three soak finds with their pre-fix tie handling through computeRadiativeTransfer: the kernel ends, and badPhotons equals
    what the instrumented instantiation dropped of the same photons."""
    n = 20000 if which == "box" else 15000
    tables = 2001 if which == "box" else 9001
    r = _run_with_test_bounds(dict(MCBRAT_TEST_LEGACY_TIES=str(ties), MCBRAT_WATCHDOG="50000", MCBRAT_MAX_EVENTS="200000"), which, seed, n, tables, 2)
    assert r["dropped_fates"] == r["bad_instrumented"], r
    assert r["bad_production"] == r["bad_instrumented"] <= 8, r
    print("production bounds, legacy ties %d seed %d: %s" % (ties, seed, r))


@pytest.mark.allow_bad_photons
@pytest.mark.timeout(900, method="thread")
@pytest.mark.parametrize("block_walk", [0, 1])
def test_production_leg_budget(block_walk):
    """The leg budget in the production instantiations (test-only build, see above): 16 legs on the conservative step cloud --
    computeRadiativeTransfer drops exactly the photons the instrumented instantiation drops, and the energy they carried is
    the only energy missing from the batch."""
    n = 20000
    r = _run_with_test_bounds(dict(MCBRAT_MAX_EVENTS="16"), "step", 0, n, 10001, block_walk)
    assert r["dropped_fates"] == r["bad_instrumented"] == r["bad_production"] > 100, r
    assert abs(r["closure"] + r["bad_production"] / n - 1.0) < 1e-3, r  # (a dropped photon had lost no weight: omega0 = 1)
