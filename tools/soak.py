#!/usr/bin/env python3
"""Soak of the default path (split pass, PROVEN stage-1 window, ties replayed on the device): random shapes, batch sizes
from one row to 1.6 M, zero and NaN rows, and - on the 768-d and 1536-d shapes - rows built against the window (residual
aligned with a hyperplane; partial sums that peak mid-way: tests/_adversary.py).  Every result is compared on the GPU with
an independent first pass: the exact-f32 kernel (a single fmaf chain per projection, proven tie window) decided by the same
replay, which the parity tests pin to the reference.   python tools/soak.py [batches [seed]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lshrs_amd import LSHHasher
from tests._adversary import adversarial_row, tent_row
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # another seed: other batches, other planted rows
rng = np.random.default_rng(2024 + seed)
shapes = [(16, 16, 768), (16, 32, 1536), (16, 16, 128), (32, 8, 96), (16, 16, 100), (16, 16, 768), (16, 16, 102), (8, 16, 768), (16, 16, 767), (20, 10, 301)]
hashers = {}
t0 = time.time(); rows = 0; bad = 0; planted = 0; worst_used = 0.0
for it in range(iters):
    nb, r, dim = shapes[int(rng.integers(0, len(shapes)))]
    n = int(rng.choice([int(rng.integers(1, 5000)), int(rng.integers(60_000, 300_000)), int(rng.integers(300_000, 1_600_000))]))
    if n * dim > 1_300_000_000: n = 1_300_000_000 // dim
    key = (nb, r, dim)
    if key not in hashers:
        hashers[key] = (LSHHasher(nb, r, dim, seed=11), LSHHasher(nb, r, dim, seed=11, precision="f32"))
    a, b = hashers[key]
    x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(1_000_003 * seed + it))
    if it % 3 == 0: x *= float(2.0 ** rng.integers(-12, 13))
    if it % 5 == 0 and n > 10: x[int(rng.integers(0, n))] = 0.0
    if it % 7 == 0 and n > 10: x[int(rng.integers(0, n)), int(rng.integers(0, dim))] = float("nan")
    if dim % 32 == 0 and dim >= 768 and n > 1000:
        m = 48
        rows_at = rng.choice(n, m, replace=False)
        adv = np.stack([(adversarial_row if j % 2 else tent_row)(a.projections[j % nb][(7 * j + it) % r],
                                                                  float(rng.choice([20.0, -20.0, 3.0, -3.0])), seed=1000 * it + j + 7919 * seed)
                        for j in range(m)])
        x[torch.from_numpy(rows_at).cuda()] = torch.from_numpy(adv).cuda()
        planted += m
    flags_a = torch.zeros(n, dtype=torch.uint8, device="cuda"); flags_b = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ka = a.hash_device(x, row_flags=flags_a); sa = dict(a.last_stats)
    kb = b.hash_device(x, row_flags=flags_b)
    ok = torch.equal(ka, kb) and torch.equal(flags_a, flags_b)
    bad += (not ok); rows += n
    used = sa.get("max_dev_units", 0.0) / a.window_info["window_units_worst_case_row"] if sa.get("window") == "proven" else 0.0
    worst_used = max(worst_used, used)
    print(f"{it:3d} shape {key} n={n:8d} route={sa.get('route')} flagged={sa.get('flagged')} max_dev={sa.get('max_dev_units', 0):.1f} "
          f"relaunches={sa.get('relaunches')} {'ok' if ok else 'MISMATCH'}", flush=True)
    del x, ka, kb
print(f"soak: {iters} batches, {rows} rows ({planted} of them built against the window), {bad} mismatches, largest measured "
      f"deviation = {worst_used:.3f} of the worst-case-row window, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
