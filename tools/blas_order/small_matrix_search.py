"""OpenBLAS's SkylakeX build sends (r, n) @ (n,) with n <= 8 contiguous columns through small-matrix kernels of its own
(kernel/x86_64/sgemv_t_microk_skylakex: lda == m, m <= 8).  For every length 2 .. 8 and every class of rows this finds, by
masking probes (fprev.py), the summation tree, and then - by trying every placement of fused multiply-adds on that tree against
NumPy bit for bit - the arithmetic.  Run under OPENBLAS_CORETYPE=SkylakeX:

    OPENBLAS_CORETYPE=SkylakeX python tools/blas_order/small_matrix_search.py
"""
import ctypes, itertools, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fprev

libm = ctypes.CDLL("libm.so.6")
libm.fmaf.restype = ctypes.c_float
libm.fmaf.argtypes = [ctypes.c_float] * 3
f32 = np.float32


def variants(t):
    """All programs for tree t: a program is a nested tuple ('leaf', k) | ('add', pL, pR) | ('fma', k, p) (= fma(a_k, x_k, p))."""
    if not isinstance(t, tuple):
        return [("leaf", t)]
    assert len(t) == 2, t
    L, R = t
    out = [("add", pl, pr) for pl in variants(L) for pr in variants(R)]
    if not isinstance(R, tuple):
        out += [("fma", R, pl) for pl in variants(L)]
    if not isinstance(L, tuple):
        out += [("fma", L, pr) for pr in variants(R)]
    return out


def run(p, a, x):
    if p[0] == "leaf":
        return f32(a[p[1]] * x[p[1]])
    if p[0] == "add":
        return f32(run(p[1], a, x) + run(p[2], a, x))
    return f32(libm.fmaf(float(a[p[1]]), float(x[p[1]]), float(run(p[2], a, x))))


def show(p):
    if p[0] == "leaf":
        return f"p{p[1]}"
    if p[0] == "add":
        return f"({show(p[1])} + {show(p[2])})"
    return f"fma({p[1]}, {show(p[2])})"


def block_type(r, row):
    """Which of the small-matrix kernels takes row `row` of an r-row band: blocks of eight rows first, then one of four, a pair, one."""
    b8 = r // 8 * 8
    if row < b8:
        return "B8"
    rem = r - b8
    if rem >= 4 and row < b8 + 4:
        return "B4"
    base = b8 + (4 if rem >= 4 else 0)
    rem -= 4 if rem >= 4 else 0
    if rem >= 2 and row < base + 2:
        return "B2"
    return "B1"


def search_row(n, r, row, rng, trials=40):
    x1 = np.ones(n, dtype=f32)
    def call(a):
        P = np.ones((r, n), dtype=f32); P[row] = a
        return (P @ x1)[row]
    tree = fprev.build(range(n), fprev.lca_sizes(n, call))
    alive = variants(tree)
    for _ in range(trials):
        P = rng.standard_normal((r, n)).astype(f32)
        x = rng.standard_normal(n).astype(f32)
        if rng.integers(0, 2):
            p64 = P[row].astype(np.float64)
            x = (x - (x @ p64) / (p64 @ p64) * p64).astype(f32)
        want = (P @ x)[row]
        alive = [p for p in alive if run(p, P[row], x).view(np.uint32) == want.view(np.uint32)]
        if not alive:
            break
    return tree, alive


def main():
    rng = np.random.default_rng(3)
    table = {}
    for n in range(1, 9):
        for r in range(2, 26):
            for row in range(r):
                tree, alive = search_row(n, r, row, rng) if n > 1 else (0, [("leaf", 0)])
                key = (n, block_type(r, row))
                got = tuple(show(p) for p in alive)
                table.setdefault(key, {}).setdefault(got, []).append((r, row))
    for key in sorted(table):
        for got, where in table[key].items():
            print(f"n={key[0]} {key[1]}: {got if got else 'NOTHING MATCHES'}  <- {len(where)} (r, row) pairs, e.g. {where[:4]}", flush=True)


if __name__ == "__main__":
    main()
