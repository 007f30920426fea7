"""cProfile of `LSHHasher.hash_vector` and `LSHRS.ingest` (one vector per call: the reference's own calling pattern)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lshrs_amd import LSHRS, InMemoryStorage, LSHHasher

h = LSHHasher(16, 16, 768, seed=42)
xs = np.random.default_rng(3).standard_normal((4096, 768)).astype(np.float32)
for i in range(200):
    h.hash_vector(xs[i])
t0 = time.perf_counter()
for i in range(2000):
    h.hash_vector(xs[i])
print(f"hash_vector: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per call")
pr = cProfile.Profile(); pr.enable()
for i in range(2000):
    h.hash_vector(xs[i])
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
idx = LSHRS(dim=768, num_perm=256, storage=InMemoryStorage())
for i in range(200):
    idx.ingest(i, xs[i])
t0 = time.perf_counter()
for i in range(2000):
    idx.ingest(1000 + i, xs[i])
print(f"ingest: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per call")
