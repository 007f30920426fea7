#!/usr/bin/env python3
"""Soak of the proven windows ACROSS SHAPES: dimensions 32 .. 8192 (the split pass's whole range) and band layouts with one
to three column blocks, padded bands, 128 key columns; per shape Gaussian rows, rows built against the window
(tests/_adversary.py: residual-aligned and tent rows, for hyperplanes drawn at random), rows at extreme and mixed scales,
rows dominated by one element, rows with empty k-tiles.  Every batch: split pass + proven window against the exact-f32
kernel + proven tie window (same replay) on the whole batch, and against the reference-literal NumPy loop on a slice.
    python tools/soak_shapes.py [rounds] [new | r6]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from lshrs_amd import LSHHasher
from oracle.lshrs_oracle import hash_batch_literal_packed
from tests._adversary import adversarial_row, tent_row

SHAPES = ((16, 16, 32), (16, 16, 64), (16, 16, 96), (32, 8, 128), (32, 8, 256), (8, 16, 384), (16, 4, 768), (4, 64, 512),
          (12, 24, 768), (20, 13, 640), (64, 8, 768), (16, 16, 1024), (16, 32, 2048), (8, 32, 4096), (16, 16, 8192),
          # bands the host BLAS does not take four rows at a time (its unfused kernels for the rest), vectors of several blocks
          (20, 10, 768), (40, 5, 768), (8, 25, 768), (12, 23, 512), (16, 18, 4128), (20, 6, 128), (10, 10, 768), (20, 5, 384),
          # key rows that are not whole 32-bit words (stage 2's atomics straddle rows)
          (25, 8, 768), (5, 20, 768), (10, 20, 512), (5, 11, 96), (3, 5, 64), (7, 9, 1024),
          # rows that are not whole k-tiles; 8 m + 4 elements (the library takes the first four first)
          (16, 16, 300), (20, 10, 100), (8, 7, 200), (16, 16, 1000), (4, 6, 1004), (6, 11, 36), (8, 12, 12),
          # compact column blocks in stage 1
          (128, 4, 768), (21, 12, 768), (60, 9, 384), (33, 7, 640), (100, 2, 512), (50, 5, 1536),
          # partial k-tiles in stage 1 (with a second block of the library; with compact column blocks)
          (16, 16, 5000), (40, 5, 100), (16, 16, 40),
          # round 4: short vectors on the resident-image kernel (every instantiation family: 2 / 4 / 8 k-tiles, 4 .. 16 column tiles)
          (16, 4, 128), (8, 16, 128), (16, 8, 64), (16, 8, 256), (5, 12, 200), (128, 2, 128), (24, 8, 96), (16, 16, 128), (2, 2, 12),
          # round 4: vector lengths that are not a multiple of four (the host library's scalar tail: blas model 1 / 2)
          (16, 16, 102), (5, 8, 30), (20, 10, 301), (4, 13, 1001), (8, 7, 99), (2, 16, 4099), (16, 16, 767), (3, 2, 9), (16, 4, 129),
          # round 4: one row per band (the host's sdot)
          (64, 1, 64), (16, 1, 128), (32, 1, 768),
          # round 5: one row per band at lengths WITH a tail behind the last whole 32 (summed in a double), down to two elements
          (64, 1, 100), (16, 1, 33), (40, 1, 31), (24, 1, 7), (8, 1, 2), (128, 1, 770), (200, 1, 96), (32, 1, 1000), (16, 1, 4100),
          # round 5: long rows - stage 2 column by column (the list sorted by key column; dim >= 1024), incl. bands of any height,
          # partial k-tiles, several blocks of the library
          (16, 32, 1536), (25, 8, 1024), (20, 10, 1100), (8, 25, 4096), (12, 7, 2052), (40, 5, 1280),
          # round 5: a scalar tail through the split pass (both stage-1 kernels, the bucket stage 2, the zero-padded 256-column image)
          (16, 32, 1537), (25, 8, 1023), (8, 16, 771), (20, 6, 127), (16, 16, 33), (12, 13, 2050),
          # round 5: 8 m + 4 elements beyond 4096 (the library's short last block)
          (8, 6, 4100), (4, 16, 8196), (16, 16, 4109),
          # round 5: fewer than 9 elements with bands of two rows and more (model 2 on the Haswell / Zen build, 3 on the SkylakeX build)
          (8, 4, 2), (5, 3, 7), (6, 2, 8), (4, 13, 8), (16, 16, 4), (3, 9, 5), (7, 6, 3), (2, 33, 6),
          # round 6: sig16_kernel's two workgroup shapes at odd and even k-tile counts (5 .. 48), one and two column blocks, compact
          # blocks (`r6`: each of them at a batch of ~24 000 rows - 128-row workgroups at every length - and at the usual size)
          (16, 16, 160), (16, 16, 224), (16, 16, 288), (16, 16, 348), (32, 16, 224), (16, 16, 416), (32, 16, 428), (20, 10, 429),
          (16, 16, 520), (16, 16, 768), (32, 16, 1536), (128, 4, 768), (16, 16, 353), (40, 5, 300))


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    only_new = len(sys.argv) > 2 and sys.argv[2] == "new"          # (the shapes round 5 added)
    r6 = len(sys.argv) > 2 and sys.argv[2] == "r6"                 # (the shapes round 6 added, at two batch sizes)
    shapes = SHAPES[SHAPES.index((64, 1, 100)):] if only_new else SHAPES
    if r6:
        shapes = tuple(sh + (small,) for sh in SHAPES[SHAPES.index((16, 16, 160)):] for small in (True, False))
    t0 = time.time()
    rows = bad = batches = audit_bad = audited = 0
    routes = {}
    worst = 0.0
    for rnd in range(rounds):
        for shape in shapes:
            nb, r, dim = shape[:3]
            rng = np.random.default_rng(7919 * rnd + 31 * dim + nb)
            a = LSHHasher(nb, r, dim, seed=3 + rnd)
            b = LSHHasher(nb, r, dim, seed=3 + rnd, precision="f32")
            n = int(np.clip((24 << 20) // dim, 20_000, 200_000)) + int(rng.integers(0, 300))
            if len(shape) > 3:
                n = 24_000 + int(rng.integers(0, 300)) if shape[3] else max(n, 70_000)
            g = torch.Generator("cuda").manual_seed(17 * rnd + dim + nb)
            x0 = torch.randn(n, dim, device="cuda", generator=g)
            planes = np.concatenate([np.asarray(p, dtype=np.float32) for p in a.projections])
            built = []
            for i in range(384):                       # rows against the window, each for a hyperplane of its own
                p = planes[int(rng.integers(0, planes.shape[0]))]
                make = adversarial_row if i % 3 else tent_row
                built.append(make(p, target_units=float(rng.uniform(-40.0, 40.0)), seed=int(rng.integers(1 << 30))))
            built = torch.from_numpy(np.stack(built)).cuda()
            reps = (n + built.shape[0] - 1) // built.shape[0]
            row_scale = torch.exp2(torch.randint(-30, 31, (n, 1), device="cuda", generator=g).float())
            outlier = x0.clone()
            outlier[torch.arange(n, device="cuda"), torch.randint(0, dim, (n,), device="cuda", generator=g)] *= 4096.0
            holes = x0.clone()
            for t in range(0, dim // 32, 2):
                holes[:, 32 * t:32 * t + 32] = 0.0
            shifted = torch.zeros(n * dim + 8, device="cuda")          # the same rows 4 bytes past a 16-byte boundary
            off_view = shifted[1:1 + n * dim].view(n, dim)
            off_view.copy_(x0)
            data = {
                "gaussian": x0,
                "offset_view": off_view,
                "adversary": built.repeat(reps, 1)[:n] * torch.exp2(torch.randint(-3, 4, (n, 1), device="cuda", generator=g).float()),
                "tiny_2^-30": x0 * 2.0 ** -30,
                "huge_2^30": x0 * 2.0 ** 30,
                "mixed_scales": x0 * row_scale,
                "one_outlier": outlier,
                "empty_k_tiles": holes,
            }
            named = None
            model = a._replay_model()
            if model in (1, 2):          # the keys pinned to the build this host runs: must be the default hasher's
                from lshrs_amd import _hostblas
                build = "openblas-skylakex" if model == 1 else "openblas-haswell"
                if _hostblas.named_model(build, r, dim) == model or (dim % 4 == 0 and r > 1 and _hostblas.named_model(build, r, dim)):
                    named = LSHHasher(nb, r, dim, seed=3 + rnd, reference_blas=build)
            for dname, x in data.items():
                ka = a.hash_device(x)
                sa = dict(a.last_stats)
                kb = b.hash_device(x)
                ok = torch.equal(ka, kb) and (named is None or torch.equal(named.hash_device(x), ka))
                sl = slice(n // 3, n // 3 + (400 if dim % 4 == 0 and dname != "offset_view" else 2000))
                routes[sa.get("route")] = routes.get(sa.get("route"), 0) + 1
                audit_bad += int(sa.get("audit_sign_disagreements", 0))
                audited += int(sa.get("audited_unflagged", 0))
                ok_ref = np.array_equal(ka[sl].cpu().numpy(), hash_batch_literal_packed(a.projections, x[sl].cpu().numpy()))
                used = sa.get("max_dev_units", 0.0) / a.window_info.get("window_units_worst_case_row", float("inf"))
                worst = max(worst, used)
                bad += (not ok) or (not ok_ref)
                rows += n
                batches += 1
                print(f"round {rnd} {nb:2d}x{r:2d} dim {dim:5d} {dname:14s} n={n:6d} route={sa.get('route')} window={a.tau1_ulps:7.1f} "
                      f"flagged={sa.get('flagged')} flips={sa.get('sign_flips')} max_dev={sa.get('max_dev_units', 0):7.1f} "
                      f"({used:.3f} of worst-case window) {'ok' if ok and ok_ref else 'MISMATCH'}", flush=True)
            a.close()
            b.close()
            if named is not None:
                named.close()
    print(f"shape soak: {batches} batches, {rows} rows, {bad} mismatches, largest measured deviation = {worst:.3f} of the "
          f"worst-case-row window, routes {routes}, {audited} un-flagged projections audited ({audit_bad} sign disagreements), "
          f"{time.time() - t0:.0f} s")
    sys.exit(1 if bad or audit_bad or routes.get("plain") else 0)


if __name__ == "__main__":
    main()
