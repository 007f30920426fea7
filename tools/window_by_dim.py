"""Largest stage-1 deviation |y1 - y_BLAS| (units of 2^-24 ||x|| ||p||, over every flagged projection) by dimension:
the data behind the default window's scaling with dim."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher
for dim in (32, 64, 96, 128, 256, 384, 768, 1024, 1536, 3072, 4096):
    n = min(600_000, 600_000_000 // dim)
    h = LSHHasher(16, 16, dim, seed=3, tau1_ulps=512.0, margin_guard=0.0, audit_every=0)
    worst = 0.0; flagged = 0
    for seed in range(3):
        x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(seed))
        if seed == 1:
            x = x / x.norm(dim=1, keepdim=True)
        if seed == 2:
            x = x * torch.exp(1.5 * torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(9)))
        h.hash_device(x)
        worst = max(worst, h.last_stats["max_dev_units"]); flagged += h.last_stats["flagged"]
    print(f"dim {dim:5d}: max deviation {worst:7.2f} units over {flagged} flagged projections; x sqrt(dim/768) = {worst * (dim / 768) ** 0.5:6.2f}", flush=True)

print("--- with the default window (64 x sqrt(768 / dim)) and the guard on: window, largest deviation, escalations")
for dim in (32, 64, 96, 128, 256, 384, 768, 1536, 3072, 4096):
    n = min(600_000, 600_000_000 // dim)
    h = LSHHasher(16, 16, dim, seed=3, audit_every=0)
    worst = 0.0
    for seed in range(3):
        x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(seed))
        if seed == 1:
            x = x / x.norm(dim=1, keepdim=True)
        if seed == 2:
            x = x * torch.exp(1.5 * torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(9)))
        h.hash_device(x)
        worst = max(worst, h.last_stats["max_dev_units"])
    print(f"dim {dim:5d}: window {h.tau1_ulps:6.1f} ({h.window_mode['tau1']}), max deviation {worst:6.2f}, escalations {h.margin_escalations}, flagged in the last batch {h.last_stats['flagged']} of {n * 256}", flush=True)
