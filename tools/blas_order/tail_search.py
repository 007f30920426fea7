"""How does this host's BLAS treat what `tools/blas_order/search.py` does not cover?  (CPU only; round 4)
  1. the `dim % 4` elements behind the last group of four in `P_band @ x` (sgemv_t's scalar tail): every order of the tail
     products, each added fused or unfused, as a chain from y or summed first - against NumPy, bit for bit;
  2. `(1, dim) @ (dim,)` - a band of ONE row, which NumPy sends to sdot: lanes x accumulators x reduction trees, for lengths
     without a tail.
Run under OPENBLAS_CORETYPE=SkylakeX / Haswell to see both builds of the library:  python tools/blas_order/tail_search.py"""
import ctypes
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from lshrs_amd import _hostblas

lib = _hostblas.load()
f = np.float32
rng = np.random.default_rng(1)


def fma(a_, b_, c_):
    return f(np.float64(a_) * np.float64(b_) + np.float64(c_))


def body_val(P, x, i, r, body):
    pb, xb = np.ascontiguousarray(P[i, :body]), np.ascontiguousarray(x[:body])
    return f(lib.lshrs_tb_model_row_dot(pb.ctypes.data, xb.ctypes.data, body, 1, i, r))


def exprs(m3):
    out = {}
    for perm in itertools.permutations(range(m3)):
        for fm in itertools.product((0, 1), repeat=m3):
            def chain_y(y, a, x, perm=perm, fm=fm):
                s = y
                for j, u in zip(perm, fm):
                    s = fma(a[j], x[j], s) if u else f(s + f(a[j] * x[j]))
                return s
            out[("chain from y", perm, fm)] = chain_y

            def chain_s(y, a, x, perm=perm, fm=fm):
                s = f(a[perm[0]] * x[perm[0]])
                for j, u in zip(perm[1:], fm[1:]):
                    s = fma(a[j], x[j], s) if u else f(s + f(a[j] * x[j]))
                return f(y + s)
            out[("tail summed first, then y + s", perm, fm[1:])] = chain_s
    return out


info = _hostblas.numpy_blas()
h = ctypes.CDLL(info[0])
core = "?"
for sym in ("scipy_openblas_get_corename64_", "openblas_get_corename64_", "openblas_get_corename"):
    if hasattr(h, sym):
        fn = getattr(h, sym)
        fn.restype = ctypes.c_char_p
        core = fn().decode()
        break
print(f"BLAS {os.path.basename(info[0])}, core {core}")
print("1. scalar tail of sgemv_t (1 = fused into an fma, 0 = product rounded on its own):")
for dim in (33, 34, 35, 102, 103, 301):
    r = 8
    P = rng.standard_normal((r, dim)).astype(f)
    body, m3 = dim & ~3, dim & 3
    E = exprs(m3)
    ok = {k: True for k in E}
    for t in range(60):
        x = rng.standard_normal(dim).astype(f)
        y = P @ x
        for i in range(r):
            yb = body_val(P, x, i, r, body)
            for k in E:
                if ok[k] and E[k](yb, P[i, body:], x[body:]).view(np.uint32) != y[i].view(np.uint32):
                    ok[k] = False
    print(f"   dim {dim} (tail of {m3}): orders that reproduce all 480 values: {[k for k in E if ok[k]][:4]}")


def acc_reduce(acc, mode):
    t = [acc[q] for q in range(acc.shape[0])]
    if mode == 0:
        v = t[0]
        for q in range(1, len(t)):
            v = (v + t[q]).astype(f)
        return v
    while len(t) > 1:
        t = [(t[2 * q] + t[2 * q + 1]).astype(f) for q in range(len(t) // 2)]
    return t[0]


def lane_reduce(v, mode):
    v = v.astype(f)
    if mode == 0:
        while len(v) > 1:
            m = len(v) // 2
            v = (v[:m] + v[m:]).astype(f)
        return v[0]
    if mode == 1:
        while len(v) > 1:
            v = (v[0::2] + v[1::2]).astype(f)
        return v[0]
    while len(v) > 4:
        m = len(v) // 2
        v = (v[:m] + v[m:]).astype(f)
    while len(v) > 1:
        v = (v[0::2] + v[1::2]).astype(f)
    return v[0]


def sim(a, x, L, U, at, lt, fold):
    A = a.reshape(-1, U, L).astype(np.float64)
    X = x.reshape(-1, U, L).astype(np.float64)
    acc = np.zeros((U, L), dtype=f)
    for i in range(A.shape[0]):
        acc = (A[i] * X[i] + acc.astype(np.float64)).astype(f)
    if fold:
        acc = (acc[:, :L // 2] + acc[:, L // 2:]).astype(f)
    return lane_reduce(acc_reduce(acc, at), lt)


print("2. a band of one row, (1, dim) @ (dim,) = sdot (lanes, accumulators, accumulator tree 0 sequential / 1 pairwise, "
      "lane tree 0 halves / 1 adjacent pairs / 2 halves to four then pairs, accumulators folded to half their lanes first):")
for dim in (128, 256, 768):
    a = rng.standard_normal((24, dim)).astype(f)
    x = rng.standard_normal((24, dim)).astype(f)
    ref = [(a[t:t + 1] @ x[t])[0] for t in range(24)]
    found = []
    for L, U, at, lt, fold in itertools.product((4, 8, 16), (1, 2, 4, 8), (0, 1), (0, 1, 2), (0, 1)):
        if dim % (L * U) or (fold and L < 8):
            continue
        if all(sim(a[t], x[t], L, U, at, lt, fold).view(np.uint32) == ref[t].view(np.uint32) for t in range(24)):
            found.append((L, U, at, lt, fold))
    print(f"   dim {dim}: {found}")
