#!/usr/bin/env python3
"""Host-resident input: NumPy in -> packed keys in NumPy out (PCIe-inclusive; never the headline value)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lshrs_amd import LSHHasher
n, dim = 1_000_000, 768
x = np.random.default_rng(1).standard_normal((n, dim), dtype=np.float32)
h = LSHHasher(16, 16, dim, seed=42)
h.hash_batch_packed(x[:300_000])
for rep in range(3):
    t = time.perf_counter(); keys = h.hash_batch_packed(x); dt = time.perf_counter() - t
    print(f"hash_batch_packed (pageable NumPy in, NumPy out): {dt*1e3:.1f} ms = {n/dt/1e6:.1f} M vec/s = {n*dim*4/dt/1e9:.1f} GB/s of input; stats {h.last_stats}", flush=True)
xp = torch.from_numpy(x).pin_memory()
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    xd = xp.to("cuda", non_blocking=True); kd = h.hash_device(xd); kh = kd.cpu(); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"pinned tensor -> device -> hash_device -> host: {dt*1e3:.1f} ms = {n/dt/1e6:.1f} M vec/s = {n*dim*4/dt/1e9:.1f} GB/s", flush=True)
from oracle.lshrs_oracle import hash_batch_literal_packed
print("parity on 3000 rows:", np.array_equal(keys[:3000], hash_batch_literal_packed(h.projections, x[:3000])))
