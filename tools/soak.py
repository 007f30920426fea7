#!/usr/bin/env python3
"""Soak test of the pipelined, split-precision default path: random batch sizes and contents, every result compared
on the GPU with the plain single-launch exact-f32 path (which the parity tests pin to the reference)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lshrs_amd import LSHHasher
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(2024)
shapes = [(16, 16, 768), (16, 32, 1536), (16, 16, 128), (32, 8, 96), (16, 16, 100)]
hashers = {}
t0 = time.time(); rows = 0; bad = 0
for it in range(iters):
    nb, r, dim = shapes[int(rng.integers(0, len(shapes)))]
    n = int(rng.choice([int(rng.integers(1, 5000)), int(rng.integers(60_000, 300_000)), int(rng.integers(300_000, 1_600_000))]))
    if n * dim > 1_300_000_000: n = 1_300_000_000 // dim
    key = (nb, r, dim)
    if key not in hashers:
        a = LSHHasher(nb, r, dim, seed=11)
        b = LSHHasher(nb, r, dim, seed=11, precision="f32", tie_replay="off"); b.pipeline_chunk_rows = 10**9
        hashers[key] = (a, b)
    a, b = hashers[key]
    x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(it))
    if it % 5 == 0 and n > 10: x[int(rng.integers(0, n))] = 0.0
    if it % 7 == 0 and n > 10: x[int(rng.integers(0, n)), int(rng.integers(0, dim))] = float("nan")
    flags_a = torch.zeros(n, dtype=torch.uint8, device="cuda"); flags_b = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ka = a.hash_device(x, row_flags=flags_a); sa = dict(a.last_stats)
    kb = b.hash_device(x, row_flags=flags_b)
    ok = torch.equal(ka, kb) and torch.equal(flags_a, flags_b)
    bad += (not ok); rows += n
    print(f"{it:3d} shape {key} n={n:8d} split={a._split_applies(n)} ties={sa.get('tie_pairs')} relaunches={sa.get('relaunches')} {'ok' if ok else 'MISMATCH'}", flush=True)
    del x, ka, kb
print(f"soak: {iters} batches, {rows} rows, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
