#!/usr/bin/env python3
"""Stage-1 time and in-kernel clock of whatever build LSHRS_HIP_LIBRARY points at (attribution probes give wrong keys by
design: guard and audit are off here)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher
import bench
n, dim = 1_000_000, 768
x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(1000))
h = LSHHasher(16, 16, dim, seed=42, margin_guard=0.0, audit_every=0)
keys = h.hash_device(x)
for _ in range(300):
    h.hash_device(x, out=keys)
h.kernel_events = []
for _ in range(600):
    h.hash_device(x, out=keys)
ev, h.kernel_events = h.kernel_events, None
clk = bench.in_kernel_clock(torch, h, x, keys)
print(json.dumps({"lib": os.path.basename(os.environ.get("LSHRS_HIP_LIBRARY", "default")), "stage1_ms": sum(e[0] for e in ev) / len(ev),
                  "clock_GHz": clk, "stage1_Mcycles": sum(e[0] for e in ev) / len(ev) * 1e-3 * clk * 1e9 / 1e6,
                  "flagged": h.last_stats.get("flagged")}), flush=True)
