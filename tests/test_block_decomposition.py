"""mcbrat_block_decomposition (host arithmetic of the block walk, mcbrat_blockwalk.hip): the boxes must PARTITION the
grid, every cell of a box must carry the box's extinction, the box bounds of a cell must contain it, and the
"spans the periodic axis" flags must be exact.  CPU only (the function needs no device)."""
import ctypes as C

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from mcbrat3d_amd._capi import lib, ptr
from tests import cases


def decompose(ext):
    """ext[ix, iy, iz] -> (blockOf[ix, iy, iz], boxes[nBlocks, 7] = x0, x1, y0, y1, z0, z1, flags)."""
    nx, ny, nz = ext.shape
    flat = np.ascontiguousarray(ext.transpose(2, 1, 0), np.float32).reshape(-1)  # x fastest
    of = np.zeros(flat.size, np.uint16)
    rec = np.zeros(4 * flat.size, np.uint32)
    nb = C.c_int32(0)
    rc = lib().mcbrat_block_decomposition(nx, ny, nz, ptr(flat), ptr(of), ptr(rec), C.byref(nb))
    assert rc == 0
    r = rec[:4 * nb.value].reshape(-1, 4)
    boxes = np.stack([r[:, 0] & 0xFFFF, r[:, 0] >> 16, r[:, 1] & 0xFFFF, r[:, 1] >> 16, r[:, 2] & 0xFFFF, r[:, 2] >> 16, r[:, 3]], axis=1)
    return of.reshape(nz, ny, nx).transpose(2, 1, 0), boxes.astype(np.int64)


def check(ext):
    of, boxes = decompose(ext)
    nx, ny, nz = ext.shape
    covered = np.zeros(ext.shape, np.int32)
    for b, (x0, x1, y0, y1, z0, z1, flags) in enumerate(boxes):
        assert 0 <= x0 < x1 <= nx and 0 <= y0 < y1 <= ny and 0 <= z0 < z1 <= nz
        sub = ext[x0:x1, y0:y1, z0:z1].astype(np.float32)
        assert np.all(sub == sub.flat[0]), "box %d mixes extinction values" % b
        assert np.all(of[x0:x1, y0:y1, z0:z1] == b), "box %d and blockOf disagree" % b
        covered[x0:x1, y0:y1, z0:z1] += 1
        assert bool(flags & 1) == (x0 == 0 and x1 == nx) and bool(flags & 2) == (y0 == 0 and y1 == ny)
    assert np.all(covered == 1), "the boxes do not partition the grid"
    return boxes


def test_the_reference_domains():
    assert len(check(cases.step_cloud()["components"][0]["ext"])) == 2       # two slabs side by side
    b = check(cases.plane_parallel()["components"][0]["ext"])
    assert len(b) == 1 and b[0, 6] == 3                                      # one box, spans x and y
    assert len(check(cases.homog_lw(n=6)["components"][0]["ext"])) == 1
    lay = np.zeros((6, 4, 12))
    lay[:, :, 0:4], lay[:, :, 4:8], lay[:, :, 8:12] = 3.0, 20.0, 7.0
    b = check(lay)
    assert len(b) == 3 and np.all(b[:, 6] == 3)
    rng = np.random.default_rng(1)
    assert len(check(rng.random((5, 4, 3)) + 1.0)) == 60                     # no two cells alike: every cell a box


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 7), st.integers(1, 6), st.integers(1, 8), st.integers(1, 4), st.integers(0, 2 ** 31 - 1))
def test_partition_invariants_on_random_media(nx, ny, nz, nvalues, seed):
    """Random media drawn from a few extinction values (so that boxes of every shape arise), including vacuum."""
    rng = np.random.default_rng(seed)
    values = np.concatenate([[0.0], rng.random(nvalues) * 10.0])
    check(values[rng.integers(0, len(values), (nx, ny, nz))])


def test_limits():
    f = np.zeros(1, np.float32)
    of, rec, nb = np.zeros(1, np.uint16), np.zeros(4, np.uint32), C.c_int32(0)
    assert lib().mcbrat_block_decomposition(0, 1, 1, ptr(f), ptr(of), ptr(rec), C.byref(nb)) == 1
    assert lib().mcbrat_block_decomposition(70000, 1, 1, ptr(f), ptr(of), ptr(rec), C.byref(nb)) == 1
