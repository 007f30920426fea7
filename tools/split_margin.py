#!/usr/bin/env python3
"""How wide must the stage-1 window of the split-precision pass be?  For shrinking windows tau1, count the key bits
on which the bf16x3 pass (+ exact fix-up inside the window) still differs from the f32 kernel: the smallest
window with zero differences brackets the largest |y_bf16x3 - y_chain| that occurred, in units of 2^-24 ||x|| ||p||."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher

def data(kind, n, dim, gen):
    if kind == "gaussian":
        return torch.randn(n, dim, device="cuda", generator=gen)
    if kind == "uniform01":           # all-positive: sum|x p| is as large as it gets relative to |y|
        return torch.rand(n, dim, device="cuda", generator=gen)
    if kind == "sparse5":
        x = torch.randn(n, dim, device="cuda", generator=gen)
        return x * (torch.rand(n, dim, device="cuda", generator=gen) < 0.05)
    if kind == "lognormal":           # heavy tail: a few coordinates dominate each row
        return torch.exp(2.5 * torch.randn(n, dim, device="cuda", generator=gen)) * torch.sign(torch.randn(n, dim, device="cuda", generator=gen))
    if kind == "pm1":
        return torch.sign(torch.randn(n, dim, device="cuda", generator=gen))
    if kind == "halfway":             # every coordinate sits on a bf16 rounding boundary twice over (1 + 2^-8 + 2^-16 patterns)
        base = 1.0 + 2.0 ** -8 + 2.0 ** -16
        return base * torch.sign(torch.randn(n, dim, device="cuda", generator=gen)) * (1.0 + torch.randint(0, 2, (n, dim), device="cuda", generator=gen).float())
    raise ValueError(kind)

for (nb, r, dim, seed) in ((16, 16, 768, 42), (16, 32, 1536, 7)):
    n = 1_000_000 if dim == 768 else 400_000
    for kind in ("gaussian", "uniform01", "sparse5", "lognormal", "pm1", "halfway"):
        gen = torch.Generator("cuda").manual_seed(11)
        x = data(kind, n, dim, gen)
        ref = LSHHasher(nb, r, dim, seed=seed, precision="f32").hash_device(x, tie_break="none")
        line = []
        for t1 in (1, 2, 4, 8, 16, 32, 64, 128, 256):
            h = LSHHasher(nb, r, dim, seed=seed, precision="bf16x3", tau1_ulps=float(t1))
            got = h.hash_device(x, tie_break="none")
            diff = (got ^ ref)
            bits = int(torch.tensor([bin(i).count("1") for i in range(256)], device="cuda")[diff.long()].sum().item())
            line.append(f"{t1}:{bits}")
        print(f"[{nb}x{r} d={dim} n={n}] {kind:10s} differing bits of {n * nb * r:.2e} by window -> " + "  ".join(line), flush=True)
        del x
