#!/usr/bin/env python3
"""Host-fed hashing of loader-sized batches (the reference's loaders default to 10 000 rows per batch): hash_batch_packed as it
is, against the same call with the source page-locked in place for the duration of the call (what the streamed path does for
large arrays)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lshrs_amd import LSHHasher
h = LSHHasher(16, 16, 768, seed=42)
rng = np.random.default_rng(0)
for n in (2_000, 10_000, 20_000, 32_000, 40_000):
    x = rng.standard_normal((n, 768)).astype(np.float32)
    h.hash_batch_packed(x)
    t0 = time.perf_counter()
    for _ in range(20):
        k = h.hash_batch_packed(x)
    dt = (time.perf_counter() - t0) / 20
    # page-lock in place, async copy, hash, copy back
    cudart = torch.cuda.cudart()
    def pinned_call():
        ok = int(cudart.cudaHostRegister(x.ctypes.data, x.nbytes, 0)) == 0
        xd = torch.empty((n, 768), dtype=torch.float32, device="cuda")
        xd.copy_(torch.from_numpy(x), non_blocking=True)
        kd = h.hash_device(xd)
        out = kd.cpu().numpy()
        if ok:
            cudart.cudaHostUnregister(x.ctypes.data)
        return out
    k2 = pinned_call()
    t0 = time.perf_counter()
    for _ in range(20):
        k2 = pinned_call()
    dt2 = (time.perf_counter() - t0) / 20
    print(f"n={n:6d}: hash_batch_packed {1e3*dt:.3f} ms ({n/dt/1e6:.2f} M vec/s, route {h.last_stats.get('route', h.last_stats.get('path', 'streamed'))}); "
          f"registered in place {1e3*dt2:.3f} ms ({n/dt2/1e6:.2f} M vec/s); equal {np.array_equal(k, k2)}", flush=True)
