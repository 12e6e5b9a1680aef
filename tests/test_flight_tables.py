"""mcbrat_flight_tables (host arithmetic of the layer-skipping walk and the clear-air flight, mcbrat_kernels.hip): the
background is each layer's most common extinction; a brick column's range must contain every cell of its 16 columns
that differs from the background and be tight; the walk's copy must carry the sign bit exactly in the cells outside
their brick column's range, and those cells must hold the background value; the depth table must add the background
over exactly the layers some brick column leaves clear.  CPU only (the function needs no device)."""
import ctypes as C

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from mcbrat3d_amd._capi import lib, ptr
from tests import cases


def tables(ext, ze):
    """ext[ix, iy, iz] -> background[nz], lo[bx, by], hi[bx, by], walk[ix, iy, iz], depth[nz+1], flights."""
    nx, ny, nz = ext.shape
    flat = np.ascontiguousarray(ext.transpose(2, 1, 0), np.float32).reshape(-1)  # x fastest
    bg = np.zeros(nz, np.float32)
    bricks = nx % 4 == 0 and ny % 4 == 0 and 2 <= nz <= 255
    rng = np.zeros((ny // 4) * (nx // 4) if bricks else 1, np.uint16)
    walk = np.zeros(flat.size, np.float32)
    depth = np.zeros(nz + 1, np.float64)
    fl = C.c_int32(-1)
    zed = np.ascontiguousarray(ze, np.float64)
    rc = lib().mcbrat_flight_tables(nx, ny, nz, ptr(flat), ptr(zed), ptr(bg), ptr(rng) if bricks else None,
                                    ptr(walk) if bricks else None, ptr(depth), C.byref(fl))
    assert rc == 0
    if not bricks:
        return bg, None, None, None, depth, fl.value
    r = rng.reshape(ny // 4, nx // 4).T.astype(np.int64)
    return bg, r & 0xFF, r >> 8, walk.reshape(nz, ny, nx).transpose(2, 1, 0), depth, fl.value


def check(ext, ze):
    ext = ext.astype(np.float32)
    nx, ny, nz = ext.shape
    bg, lo, hi, walk, depth, flights = tables(ext, ze)
    for k in range(nz):
        vals, counts = np.unique(ext[:, :, k], return_counts=True)
        assert counts[vals == bg[k]][0] == counts.max(), "layer %d: background is not a most common value" % k
    uniform = np.array([np.all(ext[:, :, k] == bg[k]) for k in range(nz)])
    dz = np.diff(ze)
    if lo is None:
        assert flights == 0
        assert np.allclose(depth, np.concatenate([[0.0], np.cumsum(np.where(uniform, bg * dz, 0.0))]), rtol=1e-12, atol=0)
        return
    differs = ext != bg[None, None, :]
    flyable = np.zeros(nz, bool)
    any_range = False
    for bx in range(nx // 4):
        for by in range(ny // 4):
            d = differs[4 * bx:4 * bx + 4, 4 * by:4 * by + 4, :].any(axis=(0, 1))
            ks = np.nonzero(d)[0]
            if len(ks):
                assert (lo[bx, by], hi[bx, by]) == (ks.min(), ks.max() + 1), "range not tight"
                any_range = True
            else:
                assert (lo[bx, by], hi[bx, by]) == (nz, 0)
            inside = (np.arange(nz) >= lo[bx, by]) & (np.arange(nz) < hi[bx, by])
            flyable |= ~inside
            w = walk[4 * bx:4 * bx + 4, 4 * by:4 * by + 4, :]
            e = ext[4 * bx:4 * bx + 4, 4 * by:4 * by + 4, :]
            marked = np.signbit(w)
            assert np.array_equal(marked, np.broadcast_to(~inside, w.shape)), "the mark is not 'outside the range'"
            assert np.array_equal(np.abs(w), e)
            assert np.all(e[:, :, ~inside] == bg[None, None, ~inside]), "a marked cell does not hold the background"
    assert flights == int(any_range)
    # (a layer of one extinction value counts even between two cloud decks, inside every range: the runs of the
    # layer-skipping walk take their optical depth from this table)
    assert np.allclose(depth, np.concatenate([[0.0], np.cumsum(np.where(flyable | uniform, bg.astype(np.float64) * dz, 0.0))]), rtol=1e-12, atol=0)


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 4), st.integers(1, 3), st.integers(2, 9), st.integers(0, 2 ** 31 - 1), st.sampled_from([0.0, 0.2, 1.0]))
def test_random_fields(bx, by, nz, seed, fill):
    rng = np.random.default_rng(seed)
    nx, ny = 4 * bx, 4 * by
    bgp = rng.choice([0.0, 0.5, 2.0], nz)
    ext = np.broadcast_to(bgp, (nx, ny, nz)).copy()
    m = rng.random((nx, ny, nz)) < fill * rng.random()
    ext[m] = rng.choice([1.0, 3.0, 7.0, 0.0], m.sum())
    ze = np.concatenate([[0.0], np.cumsum(rng.uniform(0.01, 0.05, nz))])
    check(ext, ze)


def test_workloads_and_edge_cases():
    c = cases.landsat_like(n=32, nz=24, n_entries=4)
    tot = c["components"][0]["ext"] + c["components"][1]["ext"][None, None, :]
    check(tot, c["ze"])
    bg, lo, hi, walk, depth, flights = tables(tot.astype(np.float32), c["ze"])
    assert flights == 1 and lo.min() == 8  # (the cloud base of the generator)
    # two cloud decks with a layer of one extinction value between them, inside the range of every brick column
    e = np.zeros((4, 4, 5)); e[:, :, 0] = np.arange(16).reshape(4, 4) + 1.0; e[:, :, 4] = e[:, :, 0]; e[:, :, 2] = 3.0
    bgd, lo, hi, walk, depth, flights = tables(e.astype(np.float32), np.linspace(0.0, 1.0, 6))
    assert (lo[0, 0], hi[0, 0]) == (0, 5) and depth[3] - depth[2] == pytest.approx(3.0 * 0.2)
    check(e, np.linspace(0.0, 1.0, 6))
    # a homogeneous medium: every brick column is background throughout, no flights
    e = np.full((8, 4, 5), 2.5)
    check(e, np.linspace(0.0, 1.0, 6))
    assert tables(e.astype(np.float32), np.linspace(0.0, 1.0, 6))[5] == 0
    # column counts that are not multiples of four, a single layer: no brick columns; the depth covers the one-extinction layers
    check(np.random.default_rng(1).random((6, 4, 5)), np.linspace(0.0, 1.0, 6))
    check(np.ones((4, 4, 1)), np.array([0.0, 0.3]))
    f = np.zeros(16, np.float32)
    d = np.zeros(2)
    fl = C.c_int32(0)
    assert lib().mcbrat_flight_tables(0, 4, 1, ptr(f), ptr(d), ptr(f), None, None, ptr(d), C.byref(fl)) == 1
