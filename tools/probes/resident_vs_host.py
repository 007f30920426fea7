"""`LSHRS.index` of device-resident vectors against the same rows from host memory, over shapes (key widths 1, 2, 4 bytes), batch
sizes around the chunk seams and ids with repeats: the stores must hold the same buckets.   python tools/probes/resident_vs_host.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lshrs_amd import LSHRS, InMemoryStorage

dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
bad = 0
for num_perm, dim in ((64, 128), (256, 768), (512, 1536), (200, 100), (128, 64)):
    for n in (1, 255, 70_001, 524_288, 524_289, 1_100_000):
        if n * dim > 900_000_000:
            continue
        x = torch.randn(n, dim, device=dev, generator=torch.Generator(dev).manual_seed(n + dim))
        ids = rng.integers(0, max(2, n // 2) if n % 2 else 10 * n + 1, size=n).astype(np.int64)      # (odd n: ids repeat)
        stores = []
        for src in (x, x.cpu().numpy()):
            st = InMemoryStorage()
            LSHRS(dim=dim, num_perm=num_perm, storage=st, packed_ingest=True).index(ids, src)
            stores.append(st)
        same = stores[0].bucket_contents() == stores[1].bucket_contents()
        bad += not same
        print(f"num_perm {num_perm:3d} dim {dim:4d} n {n:8d}: buckets {'equal' if same else 'DIFFER'} ({len(stores[0].bucket_contents())} buckets)", flush=True)
        del x, stores
print("mismatches:", bad)
sys.exit(1 if bad else 0)
