#!/usr/bin/env python3
"""Sustained stage-1 / stage-2 kernel times of the default hasher on 1M x 768 (>= 2 s of back-to-back steps)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lshrs_amd import LSHHasher

n, dim = 1_000_000, 768
x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(1000))
h = LSHHasher(16, 16, dim, seed=42)
keys = h.hash_device(x)
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    h.kernel_events = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 1500
    for _ in range(steps):
        h.hash_device(x, out=keys)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev, h.kernel_events = h.kernel_events, None
    print(json.dumps({"lib": os.environ.get("LSHRS_HIP_LIBRARY", "default"), "round": rnd, "ms_per_step": 1e3 * dt / steps,
                      "stage1_ms": sum(e[0] for e in ev) / len(ev), "stage2_ms": sum(e[3] for e in ev) / len(ev)}), flush=True)
