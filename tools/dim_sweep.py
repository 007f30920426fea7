"""Rate of the default path (hash_device, 1 M rows, 16 x 16) against the vector length: where the per-tile prologue and epilogue
of stage 1 weigh.   python tools/dim_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher

n = 1_000_000
for dim in (256, 288, 300, 320, 384, 448, 512, 640, 768, 1024):
    h = LSHHasher(16, 16, dim, seed=42)
    x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(dim))
    keys = h.hash_device(x)
    for _ in range(30):
        h.hash_device(x, out=keys)
    h.kernel_events = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        h.hash_device(x, out=keys)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    ev, h.kernel_events = h.kernel_events, None
    s1 = sum(e[0] for e in ev) / max(1, len(ev)) if ev else float("nan")
    s2 = sum(e[3] for e in ev if e[3]) / max(1, len(ev)) if ev else float("nan")
    flop = 3 * 2 * dim * 256 * n
    print(f"dim {dim:5d} route {h.last_stats.get('route'):13s} {n / dt / 1e6:8.1f} M vec/s  step {dt * 1e3:.3f} ms  stage1 {s1:.3f} ms = {flop / (s1 * 1e-3) / 2.5e15:.3f} of bf16 peak  "
          f"stage2 {s2:.3f}  hbm frac {n * (4 * dim + 32) / dt / 8e12:.3f}  us per k-tile-round {s1 * 1e3 / 16 / ((dim + 31) // 32):.2f}", flush=True)
    del x, keys
