#!/usr/bin/env python3
"""How far apart are the GPU fmaf chain and this host's BLAS sgemv, in units of u*||x||*||p||?
(developer tool; supports the choice of the tie threshold tau)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lshrs_amd import LSHHasher

U = 2.0 ** -24
for (nb, r, dim, seed, n) in [(16, 16, 768, 42, 400_000), (16, 32, 1536, 7, 100_000), (16, 4, 128, 42, 400_000)]:
    h = LSHHasher(nb, r, dim, seed=seed, precision="f32")
    rng = np.random.default_rng(5)
    for kind in ("gauss", "positive", "sparse"):
        x = rng.standard_normal((n, dim)).astype(np.float32)
        if kind == "positive":
            x = np.abs(x) + 0.5            # strongly non-zero-mean embeddings
        elif kind == "sparse":
            x *= (rng.random((n, dim)) < 0.05)   # 5 % non-zeros
            x[:, 0] += 1e-3
        y_gpu = h.project_device(torch.from_numpy(x).cuda()).cpu().numpy()
        t = time.perf_counter()
        y_cpu = np.concatenate([np.matmul(p, x[:, :, None])[:, :, 0] for p in h.projections], axis=1)
        dt = time.perf_counter() - t
        p64 = np.concatenate(h.projections).astype(np.float64)
        scale = np.linalg.norm(x.astype(np.float64), axis=1)[:, None] * np.linalg.norm(p64, axis=1)[None, :] * U
        d = np.abs(y_gpu.astype(np.float64) - y_cpu.astype(np.float64)) / scale
        y64 = x.astype(np.float64) @ p64.T
        eg = np.abs(y_gpu - y64) / scale
        ec = np.abs(y_cpu - y64) / scale
        near = np.abs(y64) / scale < 64          # projections anywhere near the tie window
        flips = int(((y_gpu > 0) != (y_cpu > 0)).sum())
        worst_flip = float((np.abs(y_gpu) / scale)[(y_gpu > 0) != (y_cpu > 0)].max()) if flips else 0.0
        print(f"[{nb}x{r} d={dim} {kind:8s} n={n}] gpu-vs-blas: max {d.max():.2f} p99.999 {np.quantile(d, 0.99999):.2f} rms {np.sqrt((d**2).mean()):.3f} | "
              f"near-zero only: max {d[near].max() if near.any() else 0:.2f} ({int(near.sum())} samples) | gpu-vs-exact max {eg.max():.2f}, blas-vs-exact max {ec.max():.2f} | "
              f"sign flips {flips}, largest |y_gpu| among flips {worst_flip:.2f} u.|x||p|  (cpu {dt:.1f}s)")
