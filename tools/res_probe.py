#!/usr/bin/env python3
"""Stage-1 kernel time of the resident-image kernel (sig16r_kernel) for whatever build LSHRS_HIP_LIBRARY points at - the
attribution builds of tools/ab_build.py (-DLSHRS_AB_RES_NO_MAIN: x stream + epilogue only; -DLSHRS_AB_RES_NO_EPILOGUE: x stream
+ split + MFMAs only) give wrong keys by design: guard, audits and checks are off here."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lshrs_amd import LSHHasher

n = 1_000_000
for nb, r, dim in ((16, 4, 128), (20, 6, 128), (8, 16, 128), (16, 16, 64), (24, 8, 64), (16, 8, 256)):
    x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(dim + nb))
    h = LSHHasher(nb, r, dim, seed=42, tau1_ulps=64.0, tau_ulps=8.0, margin_guard=0.0, audit_every=0, audit_unflagged=0)
    keys = h.hash_device(x)
    for _ in range(30):
        h.hash_device(x, out=keys)
    h.kernel_events = []
    for _ in range(60):
        h.hash_device(x, out=keys)
    ev, h.kernel_events = h.kernel_events, None
    print(json.dumps({"lib": os.path.basename(os.environ.get("LSHRS_HIP_LIBRARY", "default")), "shape": f"{nb}x{r}x{dim}",
                      "stage1_us": 1e3 * sorted(e[0] for e in ev)[len(ev) // 2], "stage2_us": 1e3 * sorted(e[3] for e in ev)[len(ev) // 2],
                      "flagged": h.last_stats.get("flagged")}), flush=True)
