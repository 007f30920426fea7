"""`LSHRS.index(ids, x)` with x already on the GPU (round 6): rate by batch size, and where the host's share goes (cProfile).
    python tools/resident_ingest.py"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lshrs_amd import LSHRS, InMemoryStorage

dim = 768
dev = torch.device("cuda:0")
x = torch.randn(4_000_000, dim, device=dev, generator=torch.Generator(dev).manual_seed(5))
for n in (500_000, 1_000_000, 2_000_000, 4_000_000):
    idx = LSHRS(dim=dim, num_perm=256, storage=InMemoryStorage(record_batches=False), packed_ingest=True)
    idx.index(np.arange(200_000), x[:200_000])
    best = None
    for rep in range(2):
        ids = np.arange(n, dtype=np.int64) + 10_000_000 * (rep + 1)
        t0 = time.perf_counter()
        idx.index(ids, x[:n])
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"{n:9d} rows: {best * 1e3:7.2f} ms = {n / best / 1e6:6.1f} M vec/s", flush=True)
    del idx
idx = LSHRS(dim=dim, num_perm=256, storage=InMemoryStorage(record_batches=False), packed_ingest=True)
idx.index(np.arange(200_000), x[:200_000])
pr = cProfile.Profile(); pr.enable()
idx.index(np.arange(4_000_000, dtype=np.int64) + 10**8, x)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
