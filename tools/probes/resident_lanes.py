"""Device-resident loader batches through a hasher with several lanes (`devices=[0, 0]`: two lanes on one GPU): same buckets as one lane."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lshrs_amd import LSHRS, InMemoryStorage, LSHHasher

dev = torch.device("cuda:0")
n, dim = 600_000, 128
x = torch.randn(n, dim, device=dev, generator=torch.Generator(dev).manual_seed(9))
ids = np.arange(n, dtype=np.int64)
stores = []
for devices in (None, [0, 0], [0, 0, 0]):
    st = InMemoryStorage()
    idx = LSHRS(dim=dim, num_perm=64, storage=st, packed_ingest=True, **({"devices": devices} if devices else {}))
    idx.create_signatures(format="batches", batches=((ids[lo:lo + 150_000], x[lo:lo + 150_000]) for lo in range(0, n, 150_000)))
    stores.append(st.bucket_contents())
    print(devices, len(stores[-1]), sum(v for v, _ in st.packed_batches), flush=True)
print("equal:", all(s == stores[0] for s in stores[1:]))
sys.exit(0 if all(s == stores[0] for s in stores[1:]) else 1)
