"""Which candidate order of sdot_candidates.c reproduces NumPy's (1, n) @ (n,) bit for bit, for n = 1 .. 139 and beyond?
Run under OPENBLAS_CORETYPE=SkylakeX / Haswell: profiles/r05_blas_sdot_order.log."""
import ctypes, numpy as np, sys
import os
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libsdot_candidates.so"))   # gcc -O2 -ffp-contract=off -mfma -shared -fPIC sdot_candidates.c -o libsdot_candidates.so -lm
for f in (lib.sdot_skx, lib.sdot_hsw):
    f.restype = ctypes.c_float; f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
rng = np.random.default_rng(3)
res = {}
for n in list(range(1, 140)) + [200, 300, 500, 770, 1000, 1540]:
    ok = {k: True for k in ("skx0", "skx1", "hsw0", "hsw1")}
    for t in range(30):
        a = rng.standard_normal((1, n)).astype(np.float32)
        x = rng.standard_normal(n).astype(np.float32)
        if t % 2:
            p = a[0].astype(np.float64); x = (x - (x @ p) / (p @ p) * p).astype(np.float32) if n > 1 else x
        want = (a @ x)[0]
        for name, f, fma in (("skx0", lib.sdot_skx, 0), ("skx1", lib.sdot_skx, 1), ("hsw0", lib.sdot_hsw, 0), ("hsw1", lib.sdot_hsw, 1)):
            got = np.float32(f(a.ctypes.data, x.ctypes.data, n, fma))
            if got.view(np.uint32) != want.view(np.uint32): ok[name] = False
    res[n] = [k for k, v in ok.items() if v]
bad = {n: v for n, v in res.items() if not v}
from collections import Counter
print("match sets:", Counter(tuple(v) for v in res.values()))
print("unmatched n:", sorted(bad))
