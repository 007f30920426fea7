"""The licence of the device's tie replay (no GPU): the model of the host BLAS's summation order against this process's
NumPy.  On a host whose BLAS is of another family the recognition test reports 0 and the hasher keeps the host engine -
both outcomes are legal; what is asserted is that the check is strict and that a recognised model is exact."""

from __future__ import annotations

import numpy as np
import pytest

from lshrs_amd import LSHHasher, _hostblas


def _model_dot(a, x, model=1):
    lib = _hostblas.load()
    a = np.ascontiguousarray(a, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    return np.float32(lib.lshrs_tb_model_dot(a.ctypes.data, x.ctypes.data, a.shape[0], model))


def _model_row_dot(a, x, row, rows, model=1):
    lib = _hostblas.load()
    a = np.ascontiguousarray(a, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    return np.float32(lib.lshrs_tb_model_row_dot(a.ctypes.data, x.ctypes.data, a.shape[0], model, row, rows))


def _literal_row_dot(a, x, kind):
    """The order lshrs_host.h documents, spelled out: kind 0 eight fma chains, kind 2 the same chains with the product
    rounded on its own, kind 1 four such chains; blocks of 4096 elements, each block's sum added to y."""
    f = np.float32
    y = None
    for k0 in range(0, a.shape[0], 4096):
        ab, xb = a[k0:k0 + 4096], x[k0:k0 + 4096]
        lanes = 4 if kind == 1 else 8
        p = np.zeros(lanes, dtype=np.float32)
        head = ab.shape[0] % lanes
        for k in range(head):
            p[k] = f(ab[k] * xb[k])
        ab, xb = ab[head:], xb[head:]
        for k in range(ab.shape[0]):
            if kind == 0:           # (a double holds product + addend exactly unless their exponents are > 29 apart: not here)
                p[k % lanes] = f(np.float64(ab[k]) * np.float64(xb[k]) + np.float64(p[k % lanes]))
            else:
                p[k % lanes] = f(p[k % lanes] + f(ab[k] * xb[k]))
        if kind == 1:
            s = f(f(p[0] + p[1]) + f(p[2] + p[3]))
        else:
            q = [f(p[j] + p[j + 4]) for j in range(4)]
            s = f(f(q[0] + q[1]) + f(q[2] + q[3]))
        y = s if y is None else f(y + s)
    return y


def test_row_kinds_and_blocks_are_the_documented_order():
    assert _hostblas.blas_row_kinds(16).tolist() == [0] * 16
    assert _hostblas.blas_row_kinds(5).tolist() == [0, 0, 0, 0, 2]
    assert _hostblas.blas_row_kinds(6).tolist() == [0, 0, 0, 0, 1, 1]
    assert _hostblas.blas_row_kinds(7).tolist() == [0, 0, 0, 0, 1, 1, 2]
    assert _hostblas.blas_row_kinds(3).tolist() == [1, 1, 2]
    rng = np.random.default_rng(6)
    for dim in (64, 4096 + 64, 300, 12):                   # (300, 12: 8 m + 4 elements - the first four go first)
        a = rng.standard_normal(dim).astype(np.float32)
        x = rng.standard_normal(dim).astype(np.float32)
        for row, kind in enumerate(_hostblas.blas_row_kinds(7).tolist()):
            assert _model_row_dot(a, x, row, 7).view(np.uint32) == _literal_row_dot(a, x, kind).view(np.uint32), (dim, row)
    assert np.isnan(_model_row_dot(a, x, 7, 7)) and np.isnan(_model_row_dot(a, x, -1, 7))


def test_model_function_is_the_documented_order():
    rng = np.random.default_rng(5)
    a = rng.standard_normal(64).astype(np.float32)
    x = rng.standard_normal(64).astype(np.float32)
    p = np.zeros(8, dtype=np.float32)
    for k in range(64):                                    # eight interleaved single-rounded fma chains
        p[k % 8] = np.float32(np.float64(a[k]) * np.float64(x[k]) + np.float64(p[k % 8]))   # exact product, one rounding*
    q = [np.float32(p[j] + p[j + 4]) for j in range(4)]
    want = np.float32(np.float32(q[0] + q[1]) + np.float32(q[2] + q[3]))
    # (* a double holds product + addend of two floats exactly unless their exponents are > 29 apart: not here)
    assert _model_dot(a, x).view(np.uint32) == want.view(np.uint32)
    assert np.isnan(_model_dot(a[:6], x[:6]))              # n % 4 != 0 below 9 elements: not modelled
    # n % 4 elements behind the last group of four: the library's scalar tail, as its SkylakeX build contracts it (model 1)
    # and as its Haswell / Zen build leaves it (model 2)
    f = np.float32
    fma = lambda u, v, w: f(np.float64(u) * np.float64(v) + np.float64(w))      # noqa: E731
    for n in (61, 62, 63):
        body = n & ~3
        yb = _model_row_dot(a[:body], x[:body], 0, 4)
        t, u = a[body:n], x[body:n]
        if n - body == 1:
            want1, want2 = fma(t[0], u[0], yb), f(yb + f(t[0] * u[0]))
        elif n - body == 2:
            want1, want2 = f(yb + fma(t[0], u[0], f(t[1] * u[1]))), f(yb + f(f(t[0] * u[0]) + f(t[1] * u[1])))
        else:
            want1 = f(yb + fma(t[2], u[2], fma(t[0], u[0], f(t[1] * u[1]))))
            want2 = f(yb + f(f(f(t[0] * u[0]) + f(t[1] * u[1])) + f(t[2] * u[2])))
        assert _model_row_dot(a[:n], x[:n], 0, 4, model=1).view(np.uint32) == want1.view(np.uint32), n
        assert _model_row_dot(a[:n], x[:n], 0, 4, model=2).view(np.uint32) == want2.view(np.uint32), n
    # one row per band: NumPy calls sdot - f32 kernel over the whole 32s, the rest summed in a double (any length, both builds)
    for n in (61, 31, 5):
        for model, head in ((1, None), (2, None)):
            n1 = n & ~31
            kern = np.float32(0) if n1 == 0 else _model_row_dot(a[:n1], x[:n1], 0, 1, model=model)
            tail = 0.0
            for k in range(n1, n):
                tail += float(f(a[k] * x[k]))
            assert _model_row_dot(a[:n], x[:n], 0, 1, model=model).view(np.uint32) == f(tail + float(kern)).view(np.uint32), (n, model)
    assert np.isnan(_model_dot(a, x, model=3))             # unknown model


@pytest.mark.parametrize("nb,r,dim", [(16, 16, 768), (16, 32, 1536), (16, 4, 128), (4, 12, 32),
                                      (20, 10, 768), (20, 6, 128), (8, 25, 768), (10, 13, 640), (4, 7, 8192),
                                      (16, 16, 300), (20, 10, 100), (8, 7, 200), (4, 6, 1004), (3, 7, 12)])
def test_recognised_model_reproduces_numpy_bit_for_bit(nb, r, dim):
    planes = np.random.default_rng(9).standard_normal((nb, r, dim)).astype(np.float32)
    model = _hostblas.blas_order_model(planes)
    assert model in (0, 1)
    if model == 0:
        pytest.skip("this host's BLAS sums in an order the replay does not know: the host engine stays in charge")
    rng = np.random.default_rng(10)
    for t in range(40):                                    # fresh vectors, other bands than the recognition used
        b = int(rng.integers(0, nb))
        x = rng.standard_normal(dim).astype(np.float32)
        if t % 2:                                          # cancel against one row: only the order is left of y
            p = planes[b, t % r].astype(np.float64)
            x = (x - (x @ p) / (p @ p) * p).astype(np.float32)
        want = planes[b] @ x
        got = np.array([_model_row_dot(planes[b, i], x, i, r) for i in range(r)], dtype=np.float32)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    # the order matters on such data: a plain left-to-right f32 sum does not reproduce the library
    p = planes[0, 0].astype(np.float64)
    x = rng.standard_normal(dim)
    x = (x - (x @ p) / (p @ p) * p).astype(np.float32)
    seq = np.float32(0)
    for k in range(dim):
        seq = np.float32(seq + np.float32(planes[0, 0, k] * x[k]))
    assert (planes[0] @ x)[0].view(np.uint32) == _model_dot(planes[0, 0], x).view(np.uint32)
    assert abs(float(seq)) < 1e-3                          # (a tie-sized value either way)


def test_shapes_the_model_does_not_cover_are_refused():
    rng = np.random.default_rng(1)
    recognised = _hostblas.blas_order_model(rng.standard_normal((2, 4, 64)).astype(np.float32)) == 1
    # a short block behind full ones (8 m + 4 elements beyond 4096): modelled since round 5 - every block takes ITS first four first
    for shape in ((4, 8, 4100), (3, 5, 4108), (2, 6, 8196)):
        planes = rng.standard_normal(shape).astype(np.float32)
        model = _hostblas.blas_order_model(planes)
        assert model == (1 if recognised else 0), shape
        for _ in range(6 if model else 0):
            x = rng.standard_normal(shape[2]).astype(np.float32)
            got = np.array([_model_row_dot(planes[1, i], x, i, shape[1], model=model) for i in range(shape[1])], dtype=np.float32)
            assert np.array_equal((planes[1] @ x).view(np.uint32), got.view(np.uint32)), shape
    if _hostblas.blas_order_model(rng.standard_normal((2, 4, 64)).astype(np.float32)) == 1:
        # one row per band: NumPy's sdot, modelled for both builds of the library (1 / 2) at EVERY length (round 5: whole
        # 32-element steps through the build's SIMD kernel, the elements behind them summed in a double)
        for dim in (64, 128, 768, 100, 33, 31, 7, 1, 1000, 4100):
            planes = rng.standard_normal((6, 1, dim)).astype(np.float32)
            model = _hostblas.blas_order_model(planes)
            assert model in (1, 2), dim
            for _ in range(8):
                x = rng.standard_normal(dim).astype(np.float32)
                got = _model_row_dot(planes[3, 0], x, 0, 1, model=model)
                assert (planes[3] @ x)[0].view(np.uint32) == got.view(np.uint32), dim
    # a tail below 9 elements: the library's SkylakeX build takes small-matrix paths there (refused); its Haswell / Zen build runs
    # its usual kernels (model 2, round 5) - whichever this host has, a licensed model reproduces NumPy
    for shape in ((4, 4, 7), (3, 5, 3), (2, 9, 5), (5, 2, 6)):
        planes = rng.standard_normal(shape).astype(np.float32)
        model = _hostblas.blas_order_model(planes)
        assert model in ((2, 3) if recognised else (0,)), shape      # (round 5: both builds are modelled below 9 elements)
        for _ in range(8 if model else 0):
            x = rng.standard_normal(shape[2]).astype(np.float32)
            got = np.array([_model_row_dot(planes[0, i], x, i, shape[1], model=model) for i in range(shape[1])], dtype=np.float32)
            assert np.array_equal((planes[0] @ x).view(np.uint32), got.view(np.uint32)), shape
    # dim % 4 != 0 is modelled now (the scalar tail, model 1 or 2 by how this host's library compiles it)
    if _hostblas.blas_order_model(rng.standard_normal((2, 4, 64)).astype(np.float32)) == 1:
        for shape in ((8, 5, 102), (5, 8, 30), (3, 7, 101), (2, 16, 4099), (16, 16, 103)):
            planes = rng.standard_normal(shape).astype(np.float32)
            model = _hostblas.blas_order_model(planes)
            assert model in (1, 2), shape
            for _ in range(5):                     # ... and the model IS this process's NumPy on fresh vectors
                x = rng.standard_normal(shape[2]).astype(np.float32)
                for b in (0, shape[0] - 1):
                    got = np.array([_model_row_dot(planes[b, i], x, i, shape[1], model=model) for i in range(shape[1])])
                    assert np.array_equal((planes[b] @ x).view(np.uint32), got.view(np.uint32)), shape
        assert LSHHasher(8, 5, 102, seed=3)._replay_model() in (1, 2)
    h2 = LSHHasher(16, 16, 768, seed=3, tie_replay="off")
    assert h2.tie_replay == "off"
    with pytest.raises(ValueError):
        LSHHasher(16, 16, 768, tie_replay="maybe")


def test_hasher_pickles_with_and_without_the_newer_fields():
    import pickle

    h = LSHHasher(8, 16, 768, seed=3, tie_replay="off")
    g = pickle.loads(pickle.dumps(h))
    assert (g.tie_replay, g.pipeline_pair_head, g.replay_min_rows) == ("off", True, 256)
    assert np.array_equal(g.projections[0], h.projections[0]) and g._replay_scratch == {} and g._async_pending == []
    state = h.__getstate__()                  # a pickle written before these knobs existed
    for k in ("tie_replay", "_replay_scratch", "_async_pending", "_replay_events", "_plan_cache", "pipeline", "_pipes",
              "pipeline_pair_head", "replay_min_rows", "_split_shape_ok", "_replay_model_cache"):
        state.pop(k, None)
    old = LSHHasher.__new__(LSHHasher)
    old.__setstate__(state)
    assert (old.tie_replay, old.pipeline_pair_head, old.replay_min_rows) == ("auto", True, 256)
    assert old._replay_model() in (0, 1) and old._plan_cache == {}
