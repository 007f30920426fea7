#!/usr/bin/env python3
"""Stage-2 kernel time at the proven window and at a 64-unit one (1M x 768), for A/B builds (LSHRS_HIP_LIBRARY)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher
n, dim = 1_000_000, 768
x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(1000))
for tau1 in (None, 64.0):
    h = LSHHasher(16, 16, dim, seed=42, tau1_ulps=tau1)
    keys = h.hash_device(x)
    for _ in range(40):
        h.hash_device(x, out=keys)
    h.kernel_events = []
    for _ in range(60):
        h.hash_device(x, out=keys)
    ev, h.kernel_events = h.kernel_events, None
    print(json.dumps({"lib": os.path.basename(os.environ.get("LSHRS_HIP_LIBRARY", "default")), "window": h.window_mode["tau1"],
                      "stage1_ms": sum(e[0] for e in ev) / len(ev), "stage2_ms": sum(e[3] for e in ev) / len(ev),
                      "flagged": h.last_stats["flagged"]}), flush=True)
