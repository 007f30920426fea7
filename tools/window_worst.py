import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from lshrs_amd import LSHHasher
import test_gpu_signature as T
n, dim = 120_000, 768
h = LSHHasher(16, 16, dim, seed=42, margin_guard=0.0)
for name, x in T._structured_batches(torch, h, n, dim).items():
    h.hash_device(x); print(name, round(h.last_stats["max_dev_units"], 2), h.last_stats["flagged"])
rng = np.random.default_rng(99)
fams = {"int8_grid": rng.integers(-127, 128, size=(16, 16, dim)).astype(np.float32) / 64.0,
        "heavy_tail": (rng.standard_normal((16, 16, dim)) * np.exp(2.0 * rng.standard_normal((16, 16, dim)))).astype(np.float32),
        "rank4": (rng.standard_normal((16, 16, 4)) @ rng.standard_normal((4, dim))).astype(np.float32)}
x = torch.randn(80_000, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(31))
for name, planes in fams.items():
    hh = LSHHasher(16, 16, dim, seed=1, margin_guard=0.0); hh.projections = [planes[b].copy() for b in range(16)]
    for tag, d in (("gauss", x), ("unit", x / x.norm(dim=1, keepdim=True))):
        hh.hash_device(d); print("P", name, tag, round(hh.last_stats["max_dev_units"], 2), hh.last_stats["flagged"])
h2 = LSHHasher(16, 32, 1536, seed=7, margin_guard=0.0)
x2 = torch.randn(400_000, 1536, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
h2.hash_device(x2); print("c5 shape", round(h2.last_stats["max_dev_units"], 2), h2.last_stats["flagged"])
