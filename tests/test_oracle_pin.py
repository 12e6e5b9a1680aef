"""Pins the CPU oracle against the reference (runs without a GPU).

1. utility layer: bit-for-bit against tests/golden/ref_numeric.json, which was
   produced by the reference's own numericUtilities.f95 (tests/golden/make_ref_numeric.py);
   surface description: bit-for-bit against tests/golden/ref_surface.json, produced by the
   reference's own surfaceProperties.f95 (tests/golden/make_ref_surface.py);
2. MT19937: canonical known answers + the stream recorded in SURVEY.md section 8c;
3. the whole photon loop: the reference's own outputs recorded in SURVEY.md
   section 8c / BASELINE.md section 2 (step cloud, seed (/10,1,0/), 1e5 and 1e6 photons),
   reproduced to all six printed digits by replaying the same MT stream.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import cases

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def f32(bits):
    return np.array(bits, np.int32).view(np.float32)


@pytest.fixture(scope="module")
def ref():
    with open(os.path.join(GOLD, "ref_numeric.json")) as f:
        return json.load(f)


def test_lobatto_bit_exact(ref):
    for n, bits in ref["lobatto_mus"].items():
        mus, w = O.lobatto(int(n))
        assert np.array_equal(mus.view(np.int32), np.array(bits, np.int32)), n
        assert np.array_equal(w.view(np.int32), np.array(ref["lobatto_w"][n], np.int32)), n


def test_legendre_bit_exact(ref):
    tmus = np.array([-1.0, -0.73, -0.1, 0.0, 0.31, 0.85, 1.0], np.float32)
    for n, bits in ref["legendre"].items():
        P = O.legendre(int(n), tmus)
        want = f32(bits).reshape(len(tmus), int(n) + 1).T
        assert np.array_equal(P.view(np.int32), want.view(np.int32)), n


def test_find_index_family(ref):
    table_r = np.array([0.0, 0.1, 0.25, 0.26, 0.5, 0.51, 0.75, 0.99, 1.0], np.float32)
    table_d = table_r.astype(np.float64)
    for g, vb, ir, idd, im in ref["findindex"]:
        v = f32([vb])[0]
        assert O.find_index(v, table_r, g, "real") == ir
        assert O.find_index(float(v), table_d, g, "double") == idd
        assert O.find_index(v, table_d, g, "mixed") == im
    cdf = np.array([0.05, 0.05, 0.2, 0.45, 0.450001, 0.8, 0.95, 1.0])
    for vb, i in ref["findcdf"]:
        assert O.find_cdf_index(f32([vb])[0], cdf) == i


def test_surface_reflectance_bit_exact_against_the_reference():
    """computeSurfaceReflectance / makePeriodic / findIndex of src/surfaceProperties.f95:119-147, :211-230: 2 x 600
    positions (inside the surface, up to two periods outside it, exactly on the lower edge, on interior edges, a
    hair past the upper edge) evaluated by the reference's own module (oracle/_ref/ref_surface, generator
    tests/golden/make_ref_surface.py).  The oracle must return the same float, bit for bit; each patch carries a
    reflectance that encodes its indices, so this pins the patch lookup, not just a value."""
    with open(os.path.join(GOLD, "ref_surface.json")) as f:
        gold = json.load(f)["surfaces"]
    case = cases.step_cloud(0.99)
    total = 0
    for k, sfc in gold.items():
        xs = np.array(sfc["xedge"], np.int64).view(np.float64)
        ys = np.array(sfc["yedge"], np.int64).view(np.float64)
        refl = np.array([[(i + 10 * j) for j in range(1, sfc["numY"])] for i in range(1, sfc["numX"])], np.float32) / np.float32(100.0)
        P = cases.oracle_problem(dict(case, surface=(refl, xs, ys)))
        pts = np.array(sfc["points"], np.int64)
        x, y = pts[:, 0].view(np.float64), pts[:, 1].view(np.float64)
        want = pts[:, 2].astype(np.int32)
        got = np.array([P.surface_reflectance(a, b) for a, b in zip(x, y)], np.float32).view(np.int32)
        assert np.array_equal(got, want), (k, np.nonzero(got != want)[0][:10])
        assert len(set(want.tolist())) >= 0.75 * (sfc["numX"] - 1) * (sfc["numY"] - 1)  # (all but the narrowest patches were hit)
        total += len(want)
    assert total == 1200


def test_mt19937_known_answers():
    r = O.mt_rng(5489)  # canonical genrand_int32 seed
    first = [O.lib().orc_mt_next_u32(O.C.byref(r)) for _ in range(2)]
    assert first == [3499211612, 581869302]
    # init_by_array {0x123,0x234,0x345,0x456}: first outputs of mt19937ar.c's published test vector
    r = O.mt_rng([0x123, 0x234, 0x345, 0x456])
    assert [O.lib().orc_mt_next_u32(O.C.byref(r)) for _ in range(3)] == [1067595299, 955945823, 477289528]
    # seed=(/10,1,0/) -> first five getRandomReal of the reference
    r = O.mt_rng([10, 1, 0])
    got = O.random_reals(r, 5)
    want = np.array([0.2023857534, 0.5248060226, 0.0477472618, 0.3893984258, 0.9893438220])
    assert np.all(np.abs(got - want) < 5e-9)


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32-10
    assert O.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


REFERENCE_RUNS = {  # SURVEY.md section 8c / BASELINE.md section 2: reference Fortran, this container
    100000: (0.257067, 0.602074, 0.140887),
    1000000: (0.259176, 0.600209, 0.140603),
}


@pytest.mark.parametrize("n", [100000, 1000000])
def test_step_cloud_replays_reference(n):
    P = cases.oracle_problem(cases.step_cloud(ssa=0.99))
    assert P.grid_flags()[:2] == (True, True)
    rng = O.mt_rng([10, 1, 0])
    res = O.compute_radiative_transfer(P, O.solar_source(1.0, 0.0), rng, n)
    got = (res["meanFluxUp"], res["meanFluxDown"], res["meanFluxAbsorbed"])
    for g, w in zip(got, REFERENCE_RUNS[n]):
        assert abs(g - w) < 6e-7, (got, REFERENCE_RUNS[n])  # all six printed digits


def test_event_means_match_survey():
    # SURVEY.md section 8d (instrumented temporary copy of the reference, 2e5 photons)
    P = cases.oracle_problem(cases.step_cloud(ssa=0.99))
    n = 200000
    res = O.compute_rt(P, O.solar_source(1.0, 0.0), O.mt_rng([10, 1, 0]), n)
    c = {k: v / n for k, v in res["counters"].items()}
    assert abs(c["legs"] - 17.66) < 0.02 and abs(c["crossings"] - 45.37) < 0.1
    assert abs(c["collisions"] - 16.68) < 0.02 and abs(c["topExits"] - 0.317) < 0.002
    assert abs(c["surfaceHits"] - 0.667) < 0.002 and abs(c["rouletteKills"] - 0.016) < 0.001
