#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes: a few launches of each hot kernel on the BASELINE shapes plus a
plain device copy of known size (to calibrate FETCH_SIZE / WRITE_SIZE as the microarch guide prescribes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher
from lshrs_amd.similarity import cosine_scores_device, topk_desc_device

n, dim = 1_000_000, 768
g = torch.Generator("cuda").manual_seed(1000)
x = torch.randn(n, dim, device="cuda", generator=g)
keys = torch.empty((n, 16, 2), dtype=torch.uint8, device="cuda")
for prec in ("f32", "bf16x3"):          # the f32 kernel, and the split-precision pass (stage 1 + sig_fix_kernel)
    h = LSHHasher(16, 16, dim, seed=42, precision=prec)
    h.pipeline_chunk_rows = 10**9       # one launch per pass: per-dispatch counters cover the whole 1M rows
    for _ in range(3):
        h.hash_device(x, out=keys, tie_break="none")     # MODE 0 (f32) / MODE 1 with the stage-1 list (split)
    for _ in range(3):
        h.hash_device(x, out=keys)                       # MODE 1 (+ tie-break)
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)                                       # calibration: reads 3.072e9 B, writes 3.072e9 B
q, c = 10_000, 1_000
qrows = torch.randperm(n, device="cuda", generator=g)[:q]
queries = x[qrows] + 0.1 * torch.randn(q, dim, device="cuda", generator=g)
cidx = torch.randint(0, n, (q, c), device="cuda", generator=g)
for _ in range(3):
    s, st, qs = cosine_scores_device(x, queries, cidx)
    topk_desc_device(s, c)
torch.cuda.synchronize()
print("pmc target done")
