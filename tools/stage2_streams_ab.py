#!/usr/bin/env python3
"""Stage-2 kernel time for A/B builds that drop one of its two streams (-DLSHRS_AB_FIX_NO_X: no x rows, -DLSHRS_AB_FIX_NO_P:
no hyperplane rows; wrong keys by design - guard and audit off, a numeric window of the proven one's size)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher
n, dim = 1_000_000, 768
x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(1000))
h = LSHHasher(16, 16, dim, seed=42, tau1_ulps=339.0, tau_ulps=8.0, margin_guard=0.0, audit_every=0)
keys = h.hash_device(x)
for _ in range(40):
    h.hash_device(x, out=keys)
h.kernel_events = []
for _ in range(60):
    h.hash_device(x, out=keys)
ev, h.kernel_events = h.kernel_events, None
print(json.dumps({"lib": os.path.basename(os.environ.get("LSHRS_HIP_LIBRARY", "default")),
                  "stage2_ms": sum(e[3] for e in ev) / len(ev), "flagged": h.last_stats["flagged"]}), flush=True)
