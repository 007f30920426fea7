#!/usr/bin/env python3
"""Quick on-GPU timing of the signature kernel variants (developer tool; not the bench contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lshrs_amd import LSHHasher, _native

lib = _native.load()

def setv(v):
    lib.lshrs_debug_set_sig_waves(v[0]); lib.lshrs_debug_set_sig_pipe(v[1])

def time_variants(h, x, out, variants, rounds=6):
    res = {v: [] for v in variants}
    for v in variants:                      # warm
        setv(v); h.hash_device(x, out=out, tie_break="none")
    torch.cuda.synchronize()
    for _ in range(rounds):                 # interleaved rounds in one process
        for v in variants:
            setv(v)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); h.hash_device(x, out=out, tie_break="none"); b.record()
            torch.cuda.synchronize()
            res[v].append(a.elapsed_time(b))
    return {v: (sorted(t)[len(t)//2], min(t)) for v, t in res.items()}

for (nb, r, dim, n, seed) in [(16,16,768,1_000_000,42), (16,32,1536,1_000_000,7), (16,4,128,1_000_000,42)]:
    h = LSHHasher(nb, r, dim, seed=seed, precision="f32")
    h.pipeline_chunk_rows = 10**9   # single launch: this tool times the kernel itself
    x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    out = torch.empty((n, nb, h.band_bytes), dtype=torch.uint8, device="cuda")
    flops = 2.0*dim*nb*h.band_bytes*8*n
    for v, (med, best) in time_variants(h, x, out, [(4, 0), (4, 1), (8, 1)]).items():
        print(f"[{nb}x{r} dim={dim} n={n}] (W,ring)={v}: median {med:.3f} ms best {best:.3f} ms -> {n/med/1e3:.1f} M vec/s, {flops/med/1e9:.1f} TFLOP/s padded (of 157.3)")
    setv((4, 1))
    for _ in range(2):
        t0=time.perf_counter(); h.hash_device(x, out=out); torch.cuda.synchronize(); t1=time.perf_counter()
    print(f"    with host tie-break: {1e3*(t1-t0):.2f} ms  stats={h.last_stats}")
