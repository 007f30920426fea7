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



@pytest.mark.parametrize("threads", [1, 2, 5])
def test_host_engine_and_numpy_path_give_the_same_bytes(threads):
    """tie_threads=1 is NumPy's batched matmul; anything else is the native engine (private BLAS mappings, one per
    worker).  Both must equal `P_band @ x` bit for bit — keys AND projections."""
    from lshrs_amd import _hostblas

    nb, r, dim = 16, 16, 768
    rng = np.random.default_rng(threads)
    xrows = rng.standard_normal((300, dim)).astype(np.float32)
    m = 3000                                          # enough pairs for the engine to fan out
    inverse = rng.integers(0, 300, size=m).astype(np.int32)
    bands = np.sort(rng.integers(0, nb, size=m)).astype(np.int32)
    ref = LSHHasher(nb, r, dim, seed=3, tie_threads=1)._tie_patches(xrows, inverse, bands)
    if threads == 1:
        got = ref
    else:
        eng = _hostblas.TieBreakEngine(threads)
        try:
            assert eng.threads == threads
            planes = np.stack(LSHHasher(nb, r, dim, seed=3).projections)
            assert eng.shape_trusted(planes)
            got, y = eng.patch(planes, xrows, inverse, bands, want_y=True)
            for t in range(0, m, 97):
                want = planes[bands[t]] @ xrows[inverse[t]]
                assert np.array_equal(want.view(np.uint32), y[t].view(np.uint32))
        finally:
            eng.close()
    assert np.array_equal(got, ref)
    for t in range(0, m, 41):
        assert got[t].tobytes() == project_and_pack(np.stack(LSHHasher(nb, r, dim, seed=3).projections)[bands[t]],
                                                    xrows[inverse[t]])


def test_host_engine_rejects_bad_pairs_and_handles_strided_rows():
    from lshrs_amd import _hostblas

    eng = _hostblas.engine()
    if eng is None:
        pytest.skip("single-core host: the engine is not used")
    planes = np.random.default_rng(0).standard_normal((3, 9, 40)).astype(np.float32)
    wide = np.random.default_rng(1).standard_normal((10, 64)).astype(np.float32)
    xs = wide[:, :40]                                  # row stride 64 floats
    rows = np.array([0, 9, 4], dtype=np.int32)
    bands = np.array([0, 1, 2], dtype=np.int32)
    keys = eng.patch(planes, xs, rows, bands)
    for t in range(3):
        assert keys[t].tobytes() == project_and_pack(planes[bands[t]], np.ascontiguousarray(xs[rows[t]]))
    with pytest.raises(IndexError):
        eng.patch(planes, xs, np.array([10], dtype=np.int32), np.array([0], dtype=np.int32))
    with pytest.raises(ValueError):
        eng.patch(planes, xs, np.array([1], dtype=np.int32), np.array([3], dtype=np.int32))
    assert eng.patch(planes, xs, rows[:0], bands[:0]).shape == (0, 2)


def test_host_engine_survives_a_fork():
    """Worker threads do not cross fork(): a child that inherits a used engine must build its own, not wait on
    the parent's workers forever."""
    import os

    h = LSHHasher(4, 16, 64, seed=1, tie_threads=3)
    x = np.random.default_rng(0).standard_normal((300, 64)).astype(np.float32)
    inv = np.arange(300, dtype=np.int32)
    bands = np.sort(np.arange(300) % 4).astype(np.int32)
    want = h._tie_patches(x, inv, bands)
    pid = os.fork()
    if pid == 0:
        try:
            code = 0 if np.array_equal(h._tie_patches(x, inv, bands), want) else 3
        except BaseException:
            code = 4
        os._exit(code)
    import signal
    import time

    deadline = time.time() + 60
    while True:
        done, status = os.waitpid(pid, os.WNOHANG)
        if done:
            break
        if time.time() > deadline:
            os.kill(pid, signal.SIGKILL)
            os.waitpid(pid, 0)
            pytest.fail("forked child hung in the host engine")
        time.sleep(0.05)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0

