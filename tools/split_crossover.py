#!/usr/bin/env python3
"""Below how many rows does the exact-f32 kernel beat the split-precision pass?  (sets LSHHasher.split_min_elems)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher
for (nb, r, dim) in ((16, 16, 768), (16, 32, 1536), (16, 16, 128)):
    line = []
    for n in (4096, 8192, 16384, 24576, 32768, 49152, 65536, 98304, 131072):
        x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
        out = torch.empty((n, nb, (r + 7) // 8), dtype=torch.uint8, device="cuda")
        res = {}
        for prec in ("f32", "bf16x3"):
            h = LSHHasher(nb, r, dim, seed=42, precision=prec); h.split_min_rows = 1; h.split_min_elems = 0; h.pipeline_chunk_rows = 10**9
            for _ in range(3): h.hash_device(x, out=out, tie_break="none")
            ts = []
            for _ in range(9):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); h.hash_device(x, out=out, tie_break="none"); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
            res[prec] = sorted(ts)[4]
        line.append(f"{n}: f32 {res['f32']:.0f} / split {res['bf16x3']:.0f} us")
    print(f"[{nb}x{r} d={dim}] " + " | ".join(line), flush=True)
