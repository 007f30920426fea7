#!/usr/bin/env python3
"""Where LSHRS.index(packed_ingest=True) spends its time (host NumPy vectors -> bucket CSR in the in-memory store)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from lshrs_amd import LSHRS, InMemoryStorage
from lshrs_amd.packed_ops import _csr_host, bucket_csr

n, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000, 768
x = np.random.default_rng(0).standard_normal((n, dim)).astype(np.float32)
ids = np.arange(n, dtype=np.int64)
idx = LSHRS(dim=dim, num_perm=256, storage=InMemoryStorage(), packed_ingest=True)
h = idx._hasher
idx.index(ids[:30_000], x[:30_000])


def t(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, out


dt, (keys, flags) = t(lambda: h.hash_batch_packed(x, return_row_flags=True))
print(f"hash_batch_packed      {1e3 * dt:8.2f} ms  {n / dt / 1e6:6.2f} M vec/s  source={h.last_stats.get('source')}")
dt, _ = t(lambda: h.hash_batch_packed(x, return_row_flags=True, pin="never"))
print(f"  ... pin='never'      {1e3 * dt:8.2f} ms  {n / dt / 1e6:6.2f} M vec/s  source={h.last_stats.get('source')}")
xp = torch.from_numpy(x).pin_memory().numpy()
dt, _ = t(lambda: h.hash_batch_packed(xp, return_row_flags=True))
print(f"  ... pinned source    {1e3 * dt:8.2f} ms  {n / dt / 1e6:6.2f} M vec/s  source={h.last_stats.get('source')}")
dt, _ = t(lambda: torch.cuda.cudart().cudaHostRegister(x.ctypes.data, x.nbytes, 0) or torch.cuda.cudart().cudaHostUnregister(x.ctypes.data))
print(f"  register+unregister  {1e3 * dt:8.2f} ms")
dt, csr = t(lambda: bucket_csr(ids, keys))
print(f"bucket_csr (device)    {1e3 * dt:8.2f} ms  {len(csr)} buckets")
dt, _ = t(lambda: _csr_host(ids, keys), 1)
print(f"bucket_csr (NumPy)     {1e3 * dt:8.2f} ms")
dt, _ = t(lambda: idx.index(ids + 10 * n, x))
print(f"LSHRS.index (packed)   {1e3 * dt:8.2f} ms  {n / dt / 1e6:6.2f} M vec/s")
