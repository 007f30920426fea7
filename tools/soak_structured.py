#!/usr/bin/env python3
"""Soak of the proven default on inputs that are NOT Gaussian: hyperplane families (the seed's Gaussian draw, an int8 grid,
heavy-tailed, rank-4, sparse) x data families (unit-norm, rank-8, int8 grid, constant sign, heavy-tailed, sparse, bf16-exact,
planted at the edge of round 2's window), 768-d and 1536-d; every batch compared on the GPU with the exact-f32 kernel + proven
tie window + the same replay, and a slice of it with the reference-literal NumPy path.   python tools/soak_structured.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lshrs_amd import LSHHasher
from oracle.lshrs_oracle import hash_batch_literal_packed
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
t0 = time.time(); rows = 0; bad = 0; worst = 0.0; batches = 0
for rnd in range(rounds):
    for dim, nb, r in ((768, 16, 16), (1536, 16, 32)):
        rng = np.random.default_rng(100 * rnd + dim)
        plane_families = {
            "gaussian": None,
            "int8_grid": rng.integers(-127, 128, size=(nb, r, dim)).astype(np.float32) / 64.0,
            "heavy_tail": (rng.standard_normal((nb, r, dim)) * np.exp(2.0 * rng.standard_normal((nb, r, dim)))).astype(np.float32),
            "rank4": (rng.standard_normal((nb, r, 4)) @ rng.standard_normal((4, dim))).astype(np.float32),
            "sparse": (rng.standard_normal((nb, r, dim)) * (rng.random((nb, r, dim)) < 0.05)).astype(np.float32),
        }
        for pname, planes in plane_families.items():
            a = LSHHasher(nb, r, dim, seed=11 + rnd)
            b = LSHHasher(nb, r, dim, seed=11 + rnd, precision="f32")
            if planes is not None:
                a.projections = [planes[i].copy() for i in range(nb)]
                b.projections = [planes[i].copy() for i in range(nb)]
            n = int(rng.integers(60_000, 220_000)) * (768 // (dim // 2) if dim > 768 else 2) // 2
            g = torch.Generator("cuda").manual_seed(1000 * rnd + dim + len(pname))
            x0 = torch.randn(n, dim, device="cuda", generator=g)
            P = torch.from_numpy(np.concatenate([np.asarray(p, dtype=np.float32) for p in a.projections])).cuda()
            data = {
                "unit_norm": x0 / x0.norm(dim=1, keepdim=True),
                "rank8": torch.randn(n, 8, device="cuda", generator=g) @ torch.randn(8, dim, device="cuda", generator=g),
                "int8_grid": torch.randint(-127, 128, (n, dim), device="cuda", generator=g).float() / 127.0,
                "constant_sign": torch.rand(n, dim, device="cuda", generator=g) + 0.01,
                "heavy_tail": torch.randn(n, dim, device="cuda", generator=g) * torch.exp(2.0 * torch.randn(n, dim, device="cuda", generator=g)),
                "sparse": x0 * (torch.rand(n, dim, device="cuda", generator=g) < 0.05),
                "bf16_exact": x0.bfloat16().float(),                      # ||x_mid|| = 0: the second window factor vanishes
            }
            xd = torch.randn(n, dim, device="cuda", generator=g).double()
            cols = torch.randint(0, P.shape[0], (n,), device="cuda", generator=g)
            p = P[cols].double(); pn = p.norm(dim=1).clamp_min(1e-30)
            tgt = (60.0 + 10.0 * torch.rand(n, device="cuda", generator=g).double()) * 2.0 ** -24 * xd.norm(dim=1) * pn
            tgt = tgt * torch.where(torch.rand(n, device="cuda", generator=g) < 0.5, -1.0, 1.0)
            data["planted_edge"] = (xd + ((tgt - (xd * p).sum(1)) / (pn * pn))[:, None] * p).float()
            for dname, x in data.items():
                ka = a.hash_device(x); sa = dict(a.last_stats)
                kb = b.hash_device(x)
                ok = torch.equal(ka, kb)
                sl = slice(n // 2, n // 2 + 1500)
                ok_ref = np.array_equal(ka[sl].cpu().numpy(), hash_batch_literal_packed(a.projections, x[sl].cpu().numpy()))
                used = sa.get("max_dev_units", 0.0) / a.window_info["window_units_worst_case_row"]
                worst = max(worst, used); bad += (not ok) or (not ok_ref); rows += n; batches += 1
                print(f"round {rnd} dim {dim} planes {pname:10s} data {dname:13s} n={n:7d} window={a.tau1_ulps:7.1f} flagged={sa.get('flagged')} "
                      f"max_dev={sa.get('max_dev_units', 0):7.1f} ({used:.3f} of worst-case window) {'ok' if ok and ok_ref else 'MISMATCH'}", flush=True)
            a.close(); b.close()
print(f"structured soak: {batches} batches, {rows} rows, {bad} mismatches, largest measured deviation = {worst:.3f} of the worst-case-row window, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
