#!/usr/bin/env python3
"""Short-vector shapes (dim <= 300) through hash_device, for rocprofv3 --kernel-trace --stats / --pmc passes and for
wall-clock rates:  python3 tools/short_shapes.py [rows] [steps]
Prints one JSON line per shape: route, ms per step, vectors/s, fraction of the HBM roof (4 dim + key bytes per row at 8 TB/s),
and whether 2 000 rows agree with the oracle."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from lshrs_amd import LSHHasher
from oracle.lshrs_oracle import hash_batch_literal_packed

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
shapes = ((16, 4, 128), (20, 6, 128), (8, 16, 128), (16, 8, 256), (16, 16, 256), (16, 16, 300), (16, 16, 128), (16, 16, 102), (20, 6, 127), (16, 16, 301), (16, 16, 767), (16, 16, 768))
for nb, r, dim in shapes:
    h = LSHHasher(nb, r, dim, seed=42)
    x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(dim + nb))
    keys = h.hash_device(x)
    for _ in range(5):
        h.hash_device(x, out=keys)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        h.hash_device(x, out=keys)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = dict(h.last_stats)
    want = hash_batch_literal_packed(h.projections, x[:2000].cpu().numpy())
    row_bytes = 4 * dim + nb * h.band_bytes
    print(json.dumps({"shape": f"{nb}x{r}x{dim}", "rows": n, "route": st.get("route"), "ms_per_step": 1e3 * dt,
                      "vectors_per_s": n / dt, "frac_hbm_roof": n * row_bytes / dt / 8e12,
                      "flagged": st.get("flagged"), "tie_pairs": st.get("tie_pairs"),
                      "bit_exact_2000": bool(np.array_equal(keys[:2000].cpu().numpy(), want))}), flush=True)
    h.close()
    del x, keys
