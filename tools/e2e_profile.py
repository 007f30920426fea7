"""Where the time of LSHRS.index() goes, from host memory to buckets in the in-memory store (round 5, verdict item 7).

    python tools/e2e_profile.py [rows]
"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lshrs_amd import LSHRS, InMemoryStorage
from lshrs_amd import packed_ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
rng = np.random.default_rng(1)
host = rng.standard_normal((rows, 768), dtype=np.float32)
ids = np.arange(rows, dtype=np.int64)
idx = LSHRS(dim=768, num_perm=256, storage=InMemoryStorage(), packed_ingest=True)
idx.index(ids[:100_000], host[:100_000])
for rep in range(3):
    t0 = time.perf_counter()
    idx.index(ids + 10_000_000 * (rep + 1), host)
    dt = time.perf_counter() - t0
    print(f"index(): {dt * 1e3:.1f} ms = {rows / dt / 1e6:.2f} M vec/s", flush=True)
# the pieces
h = idx._hasher
t0 = time.perf_counter(); keys, flags = h.hash_batch_packed(host, return_row_flags=True); t1 = time.perf_counter()
print(f"hash_batch_packed (host -> host keys): {(t1 - t0) * 1e3:.1f} ms = {rows / (t1 - t0) / 1e6:.2f} M vec/s; {h.last_stats.get('source')}")
t0 = time.perf_counter(); csr = packed_ops.bucket_csr(ids, keys); t1 = time.perf_counter()
print(f"bucket_csr (host keys): {(t1 - t0) * 1e3:.1f} ms")
kd = torch.from_numpy(keys).cuda(); torch.cuda.synchronize()
t0 = time.perf_counter(); csr = packed_ops.bucket_csr(ids, kd); t1 = time.perf_counter()
print(f"bucket_csr (device keys): {(t1 - t0) * 1e3:.1f} ms; buckets {len(csr)}")
st = InMemoryStorage()
t0 = time.perf_counter(); st.batch_add_csr(csr); t1 = time.perf_counter()
print(f"InMemoryStorage.batch_add_csr: {(t1 - t0) * 1e3:.1f} ms")
pr = cProfile.Profile(); pr.enable(); idx.index(ids + 50_000_000, host); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
