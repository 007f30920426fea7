#!/usr/bin/env python3
"""f32 kernel vs split-precision (bf16x3 + exact-chain fix-up) pass: kernel-only and bit-exact rates, flag counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher
for (nb, r, dim, n, seed) in [(16, 16, 768, 1_000_000, 42), (16, 32, 1536, 1_000_000, 7)]:
    x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    out = torch.empty((n, nb, (r + 7) // 8), dtype=torch.uint8, device="cuda")
    for prec in ("f32", "bf16x3"):
        h = LSHHasher(nb, r, dim, seed=seed, precision=prec)
        h.pipeline_chunk_rows = 10**9
        for _ in range(3): h.hash_device(x, out=out, tie_break="none")
        ts = []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); h.hash_device(x, out=out, tie_break="none"); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        raw = sorted(ts)[5]
        h.pipeline_chunk_rows = 262144
        for _ in range(2): h.hash_device(x, out=out)
        t = time.perf_counter()
        for _ in range(5): h.hash_device(x, out=out)
        torch.cuda.synchronize(); e2e = (time.perf_counter() - t) / 5 * 1e3
        print(f"[{nb}x{r} d={dim} n={n}] precision={prec}: kernel-only {raw:.3f} ms = {n/raw/1e3:.0f} M vec/s "
              f"({(4*dim+nb*((r+7)//8))*n/raw/1e6:.0f} GB/s algorithmic) | bit-exact {e2e:.2f} ms = {n/e2e/1e3:.0f} M vec/s | "
              f"stats { {k: (round(v,2) if isinstance(v,float) else v) for k,v in h.last_stats.items()} }")
