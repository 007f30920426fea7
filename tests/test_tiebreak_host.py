"""Host half of the tie-break (lshrs_amd/hasher.py): entry decoding and the batched evaluation of
the reference's own expression.  CPU only (these helpers launch nothing)."""

from __future__ import annotations

import numpy as np
import pytest

from lshrs_amd import LSHHasher
from oracle.lshrs_oracle import project_and_pack


def entry(row, word, mask):
    return [(row << 16) | word, mask]


def test_tie_entries_to_band_pairs():
    h = LSHHasher(5, 12, 32, seed=1)            # 16 padded columns per band
    ent = np.array([entry(7, 0, 1), entry(7, 0, 1 << 16), entry(9, 2, (1 << 15) | (1 << 31)),
                    entry(3, 2, 1 << 20), entry(7, 0, 1 << 3)], dtype=np.int64)
    rows, bands = h._tie_pairs(ent)
    assert list(zip(rows.tolist(), bands.tolist())) == [(7, 0), (7, 1), (9, 4)]
    h = LSHHasher(2, 24, 100, seed=1)           # 24 columns per band: words straddle bands
    rows, bands = h._tie_pairs(np.array([entry(7, 0, (1 << 23) | (1 << 24)), entry(1, 1, 1 << 20)], dtype=np.int64))
    assert list(zip(rows.tolist(), bands.tolist())) == [(7, 0), (7, 1)]
    h = LSHHasher(16, 32, 64, seed=1)           # 32 columns per band: word == band
    rows, bands = h._tie_pairs(np.array([entry(5, 3, 0x80000000), entry(2**40, 15, 1), entry(5, 3, 1)], dtype=np.int64))
    assert list(zip(rows.tolist(), bands.tolist())) == [(5, 3), (2**40, 15)]
    h = LSHHasher(4, 64, 64, seed=1)            # 64 columns per band: two words per band
    rows, bands = h._tie_pairs(np.array([entry(1, 0, 1), entry(1, 1, 1), entry(1, 7, 1)], dtype=np.int64))
    assert list(zip(rows.tolist(), bands.tolist())) == [(1, 0), (1, 3)]


@pytest.mark.parametrize("nb,r,dim", [(16, 16, 768), (16, 4, 128), (16, 32, 1536), (3, 5, 4), (2, 24, 100), (5, 8, 30)])
def test_batched_patches_equal_the_reference_expression(nb, r, dim):
    """np.matmul(P_band, X[:, :, None]) must give, row for row, the bytes of `P_band @ x` etc."""
    h = LSHHasher(nb, r, dim, seed=3)
    rng = np.random.default_rng(nb * 1000 + r)
    m = 500
    xrows = rng.standard_normal((200, dim)).astype(np.float32)
    inverse = rng.integers(0, 200, size=m)
    bands = np.sort(rng.integers(0, nb, size=m)).astype(np.int32)
    patch = h._tie_patches(xrows, inverse, bands)
    assert patch.shape == (m, (r + 7) // 8)
    for t in range(m):
        want = project_and_pack(h.projections[bands[t]], xrows[inverse[t]])
        assert patch[t].tobytes() == want
    assert h._tie_patches(xrows, inverse[:0], bands[:0]).shape == (0, (r + 7) // 8)


@pytest.mark.parametrize("nb,r,dim", [(16, 16, 768), (16, 4, 128), (16, 32, 1536), (3, 5, 4), (2, 24, 100), (5, 8, 30)])
def test_native_loop_is_numpys_sgemv_bit_for_bit(nb, r, dim):
    """csrc/host_tiebreak.cpp must return the very floats `P_band @ x` returns on this host."""
    from lshrs_amd import _hostblas, _native

    lib = _native.load()
    if _hostblas.sgemv_pointer() is None:
        pytest.skip("NumPy's BLAS exposes no cblas_sgemv symbol here; the NumPy path is used instead")
    assert _hostblas.verified_for(lib, r, dim)
    h = LSHHasher(nb, r, dim, seed=9)
    rng = np.random.default_rng(nb + r + dim)
    xrows = rng.standard_normal((300, dim)).astype(np.float32)
    m = 4000
    inverse = rng.integers(0, 300, size=m)
    bands = np.sort(rng.integers(0, nb, size=m)).astype(np.int32)
    planes = [np.ascontiguousarray(p) for p in h.projections]
    patch, y = _hostblas.band_keys(lib, planes, xrows, inverse, bands, r, dim, threads=4, want_y=True)
    want = np.stack([h.projections[b] @ xrows[i] for b, i in zip(bands, inverse)])
    assert np.array_equal(y.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(patch, np.packbits(want > 0, axis=1, bitorder="little"))
    # and the hasher picks it up / agrees with its NumPy fallback
    fast = h._tie_patches(xrows, inverse, bands)
    h.native_tie_break = False
    assert np.array_equal(fast, h._tie_patches(xrows, inverse, bands))
