#!/usr/bin/env python3
"""Where does a bit-exact step spend its time on this box?  (developer tool)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lshrs_amd import LSHHasher

for (nb, r, dim, seed) in [(16, 16, 768, 42), (16, 32, 1536, 7)]:
    # (the host tie-break path with MEASURED windows is what this tool looks at: numbers, not the proven default)
    h = LSHHasher(nb, r, dim, seed=seed, tie_replay="off", tau_ulps=32.0, tau1_ulps=64.0)
    x = torch.randn(1_000_000, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    out = torch.empty((1_000_000, nb, h.band_bytes), dtype=torch.uint8, device="cuda")
    for label, chunk, tau in (("pipelined tau=32", 131072, 32.0), ("plain tau=32", 10**9, 32.0),
                              ("pipelined tau=16", 131072, 16.0), ("pipelined tau=8", 131072, 8.0),
                              ("pipelined chunk=65536", 65536, 32.0), ("pipelined chunk=262144", 262144, 32.0)):
        h.pipeline_chunk_rows = chunk; h.tau_ulps = tau
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter(); h.hash_device(x, out=out); torch.cuda.synchronize()
            dt = (time.perf_counter() - t) * 1e3
        print(f"[{nb}x{r} d={dim}] {label}: {dt:.2f} ms  stats={ {k: (round(v, 2) if isinstance(v, float) else v) for k, v in h.last_stats.items()} }")
