#!/usr/bin/env python3
"""Kernel time vs batch size for the two geometries (NT=8 workgroups vs fine NT=1): where is the crossover?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher, _native
lib = _native.load()
for (nb, r, dim) in [(16, 16, 768), (16, 32, 1536)]:
    h = LSHHasher(nb, r, dim, precision="f32"); h.pipeline_chunk_rows = 10**9
    xall = torch.randn(1_000_000 if dim == 768 else 400_000, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    for n in (1, 32, 128, 1024, 4096, 8192, 16384, 24576, 32768, 36864, 40960, 49152, 57344, 65536, 16960 + 983040 if dim == 768 else 131072 + 20000):
        x = xall[:n]; out = torch.empty((n, nb, h.band_bytes), dtype=torch.uint8, device="cuda")
        res = {}
        for mode in (0, 2, 1):
            lib.lshrs_debug_set_sig_fine(mode)
            for _ in range(3): h.hash_device(x, out=out, tie_break="none")
            ts = []
            for _ in range(7):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); h.hash_device(x, out=out, tie_break="none"); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            res[mode] = sorted(ts)[3]
        print(f"[{nb}x{r} d={dim}] n={n:8d}: main-only {res[0]:9.1f} us | fine-always {res[2]:9.1f} us | auto {res[1]:9.1f} us")
    lib.lshrs_debug_set_sig_fine(1)
