#!/usr/bin/env python3
"""Workload for rocprofv3 --kernel-trace / --pmc passes: a few launches of each hot kernel on the BASELINE shapes plus
a plain device copy of known size (to calibrate FETCH_SIZE / WRITE_SIZE as the microarch guide prescribes).
    python3 tools/pmc_target.py [c2|c5|all]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lshrs_amd import LSHHasher
from lshrs_amd.similarity import cosine_scores_device, topk_desc_device

what = sys.argv[1] if len(sys.argv) > 1 else "all"
g = torch.Generator("cuda").manual_seed(1000)
if what in ("c2", "all"):
    n, dim = 1_000_000, 768
    x = torch.randn(n, dim, device="cuda", generator=g)
    keys = torch.empty((n, 16, 2), dtype=torch.uint8, device="cuda")
    for kw in ({}, {"tau1_ulps": 64.0, "tau_ulps": 8.0}, {"precision": "f32"}):   # default (proven window), round 2's measured window, f32 kernel
        h = LSHHasher(16, 16, dim, seed=42, **kw)
        for _ in range(6):
            h.hash_device(x, out=keys)
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)                                       # calibration: reads 3.072e9 B, writes 3.072e9 B
    del y
    q, c = 10_000, 1_000
    qrows = torch.randperm(n, device="cuda", generator=g)[:q]
    queries = x[qrows] + 0.1 * torch.randn(q, dim, device="cuda", generator=g)
    cidx = torch.randint(0, n, (q, c), device="cuda", generator=g)
    for _ in range(3):
        s, st, qs = cosine_scores_device(x, queries, cidx)
        topk_desc_device(s, c)
    del x, keys
if what in ("c5", "all"):
    n, dim = 2_500_000, 1536                            # config 5's shape (half its rows: the per-row figures are what counts)
    x = torch.empty((n, dim), device="cuda")
    for lo in range(0, n, 500_000):
        x[lo:lo + 500_000].normal_(generator=g)
    keys = torch.empty((n, 16, 4), dtype=torch.uint8, device="cuda")
    h = LSHHasher(16, 32, dim, seed=7)
    for _ in range(4):
        h.hash_device(x, out=keys)
    y = torch.empty_like(x)
    for _ in range(2):
        y.copy_(x)
torch.cuda.synchronize()
print("pmc target done")
