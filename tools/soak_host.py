#!/usr/bin/env python3
"""Soak of the host-facing paths: hash_batch_packed (small path, plain path, streamed path with random chunk sizes) against
hash_device on the same rows, and LSHRS.index(packed_ingest=True) against the op-tuple path (bucket contents), random
sizes and shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lshrs_amd import LSHHasher, LSHRS, InMemoryStorage
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(77)
shapes = [(16, 16, 768), (16, 4, 128), (8, 16, 768), (16, 32, 1536), (4, 12, 32), (16, 16, 102), (16, 16, 767)]
bad = 0; rows = 0; t0 = time.time()
hashers = {}
for it in range(iters):
    nb, r, dim = shapes[int(rng.integers(0, len(shapes)))]
    n = int(rng.choice([int(rng.integers(1, 200)), int(rng.integers(200, 40_000)), int(rng.integers(40_000, 400_000))]))
    n = min(n, 300_000_000 // dim)
    h = hashers.setdefault((nb, r, dim), LSHHasher(nb, r, dim, seed=5))
    x = rng.standard_normal((n, dim)).astype(np.float32)
    if n > 3:
        x[int(rng.integers(0, n))] = 0.0
        x[int(rng.integers(0, n)), int(rng.integers(0, dim))] = np.nan
    want = h.hash_device(torch.from_numpy(x).cuda())
    wf = torch.zeros(n, dtype=torch.uint8, device="cuda"); h.hash_device(torch.from_numpy(x).cuda(), row_flags=wf)
    chunk = int(rng.choice([16_384, 20_000, 65_536, 131_072]))
    keys, flags = h.hash_batch_packed(x, return_row_flags=True, chunk_rows=chunk, pin=str(rng.choice(["auto", "never"])))
    ok = np.array_equal(keys, want.cpu().numpy()) and np.array_equal(flags, wf.cpu().numpy())
    if n <= 3000 and not np.isnan(x).any():
        pass
    m = min(n, 3000)
    xs = np.nan_to_num(x[:m])
    xs = xs + (np.abs(xs).sum(1, keepdims=True) == 0)      # no zero / NaN rows for index() (the NaN may sit in the zero row)
    a, b = InMemoryStorage(), InMemoryStorage()
    ids = rng.permutation(10**6)[:m].astype(np.int64)
    LSHRS(dim=dim, num_bands=nb, rows_per_band=r, num_perm=nb * r, storage=a, hasher=h, packed_ingest=True).index(ids, xs)
    LSHRS(dim=dim, num_bands=nb, rows_per_band=r, num_perm=nb * r, storage=b, hasher=h).index(ids.tolist(), xs)
    ok = ok and a.bucket_contents() == b.bucket_contents()
    bad += not ok; rows += n
    print(f"{it:3d} shape {(nb, r, dim)} n={n:7d} chunk={chunk:6d} {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"soak_host: {iters} batches, {rows} rows, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
