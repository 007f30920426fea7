#!/usr/bin/env python3
"""Quick on-GPU timing of the signature kernel (developer tool; not the bench contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lshrs_amd import LSHHasher

def timeit(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts)//2], ts[0]

for (nb, r, dim, n, seed) in [(16,16,768,1_000_000,42), (16,32,1536,1_000_000,7), (16,4,128,1_000_000,42)]:
    h = LSHHasher(nb, r, dim, seed=seed)
    x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    out = torch.empty((n, nb, h.band_bytes), dtype=torch.uint8, device="cuda")
    med, best = timeit(lambda: h.hash_device(x, out=out, tie_break="none"))
    flops = 2.0*dim*nb*r*n
    print(f"[{nb}x{r} dim={dim} n={n}] raw kernel: median {med:.3f} ms best {best:.3f} ms -> {n/med/1e3:.1f} M vec/s, {flops/med/1e9:.1f} TFLOP/s (of 155)")
    t0=time.perf_counter(); h.hash_device(x, out=out); torch.cuda.synchronize(); t1=time.perf_counter()
    t0=time.perf_counter(); h.hash_device(x, out=out); torch.cuda.synchronize(); t1=time.perf_counter()
    print(f"    with host tie-break: {1e3*(t1-t0):.2f} ms  stats={h.last_stats}")
