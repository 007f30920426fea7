"""In which order does this host's BLAS sum a dot product?  Tries a family of candidate orders (SIMD lanes x independent
accumulators x fma or not x two reduction trees for the accumulators x three for the lanes x K-blocking; model.c) against
what NumPy returns for `P @ x` - the reference's call, lshrs/hash/lsh.py:200 - bit for bit, row by row, and prints the
candidates that reproduce every trial.  On the hosts of round 1 (Intel Xeon and AMD EPYC 9575F, OpenBLAS 0.3.29 "SkylakeX")
exactly one order survives: 4 lanes x 2 accumulators, fma, adjacent-pair reduction = eight interleaved chains over
k mod 8 reduced ((p0+p4)+(p1+p5))+((p2+p6)+(p3+p7)).  That order is `lshrs_tb_model_dot(model=1)` and what
sig_fix8_kernel<true> replays.   build: gcc -O2 -ffp-contract=off -shared -fPIC model.c -o libmodel.so -lm"""
import ctypes, itertools, os, sys
import numpy as np
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmodel.so"))
lib.model_dot.restype = ctypes.c_float
lib.model_dot.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int]*7
rng = np.random.default_rng(1)
def trial(r, dim, ntr=40):
    P = rng.standard_normal((r, dim)).astype(np.float32)
    X = rng.standard_normal((ntr, dim)).astype(np.float32)
    ref = np.stack([P @ X[t] for t in range(ntr)])      # the reference's call: sgemv per vector
    found = []
    for L, U, fma, at, lt, blk in itertools.product((4, 8, 16, 32), (1, 2, 4, 8), (1, 0), (0, 1), (0, 1, 2), (0, 128, 256, 512, 1024)):
        if L * U > 256: continue
        ok_rows = []
        for i in range(r):
            good = True
            for t in range(ntr):
                v = lib.model_dot(P[i].ctypes.data, X[t].ctypes.data, dim, L, U, fma, at, lt, blk)
                if np.float32(v).view(np.uint32) != ref[t, i].view(np.uint32):
                    good = False; break
            ok_rows.append(good)
        if any(ok_rows):
            found.append(((L, U, fma, at, lt, blk), ok_rows))
    return found
for r, dim in ((16, 768), (32, 1536), (4, 128), (5, 100)):
    f = trial(r, dim)
    print(r, dim, "models matching some rows:", len(f))
    for m, rows in f[:12]:
        print("   ", m, "rows ok:", "".join("1" if v else "0" for v in rows))
