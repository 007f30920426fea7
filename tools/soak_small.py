#!/usr/bin/env python3
"""Soak of the small-batch replay path (lshrs_sig_hash_small_replay_f32): random batch sizes 1..128, both return forms and
both placements of x, against the reference-literal NumPy loop on this host (oracle), near-zero rows included."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lshrs_amd import LSHHasher
from oracle.lshrs_oracle import hash_batch_literal_packed

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = np.random.default_rng(4242)
bad = rows = 0
t0 = time.time()
for nb, r, dim in ((16, 16, 768), (16, 32, 1536), (16, 4, 128), (8, 16, 1024), (3, 8, 64)):
    h = LSHHasher(nb, r, dim, seed=int(rng.integers(1, 1000)))
    planes = np.concatenate([np.asarray(p) for p in h.projections])
    ties = 0
    for c in range(calls):
        n = int(rng.integers(1, 129)) if c % 3 else 1
        x = rng.standard_normal((n, dim)).astype(np.float32)
        if c % 7 == 0:                       # rows in the null space of a few hyperplanes, up to rounding: |y| ~ 1e-7 ||x|| ||p||
            k = int(rng.integers(0, n))
            for j in rng.integers(0, planes.shape[0], size=3):
                p = planes[j].astype(np.float64)
                x[k] = (x[k].astype(np.float64) - (x[k].astype(np.float64) @ p) / (p @ p) * p).astype(np.float32)
        keys = h.hash_batch_packed(x)
        ties += h.last_stats.get("tie_entries", 0)
        assert h.last_stats.get("path") == "small-replay", h.last_stats
        ok = np.array_equal(keys, hash_batch_literal_packed(h.projections, x))
        bad += not ok
        rows += n
    print(f"shape {(nb, r, dim)}: {calls} calls, projections inside the tie window: {ties}, mismatching calls so far: {bad}", flush=True)
    h.close()
print(f"soak_small: {rows} rows, {bad} mismatching calls, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
