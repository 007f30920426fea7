"""Latency of the one-vector / small-batch path by batch size, beside the reference-literal CPU loop.

    python tools/small_latency.py            # p50 / p95 in microseconds for n = 1 .. 128: both return forms, both x placements
"""

from __future__ import annotations

import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pcts(fn, reps):
    ts = []
    for i in range(reps):
        t0 = time.perf_counter()
        fn(i)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return 1e6 * ts[len(ts) // 2], 1e6 * ts[int(len(ts) * 0.95)]


def main() -> None:
    import torch

    from lshrs_amd import LSHHasher
    from oracle.lshrs_oracle import hash_vector_literal

    assert torch.cuda.is_available()
    for nb, r, dim in ((16, 16, 768), (16, 32, 1536), (16, 4, 128)):
        h = LSHHasher(nb, r, dim, seed=42)
        xs = np.random.default_rng(3).standard_normal((4096, dim)).astype(np.float32)
        h.hash_batch_packed(xs[:4])
        print(f"--- {nb} x {r}, dim {dim}", flush=True)
        cpu = pcts(lambda i: hash_vector_literal(h.projections, xs[i], dim), 200)
        print(f"reference-literal hash_vector on this host: p50 {cpu[0]:.1f} us  p95 {cpu[1]:.1f} us", flush=True)
        hv = pcts(lambda i: h.hash_vector(xs[i]), 400)
        print(f"hash_vector (GPU): p50 {hv[0]:.1f} us  p95 {hv[1]:.1f} us", flush=True)
        for n in (1, 2, 4, 8, 16, 32, 64, 128):
            row = [f"n = {n:4d}"]
            for label, poll, direct in (("poll/pinned-x", 1 << 30, 1 << 40), ("poll/device-x", 1 << 30, 0),
                                        ("copy/pinned-x", 0, 1 << 40), ("copy/device-x", 0, 0), ("shipped", None, None)):
                if poll is None:
                    del h._small_poll_bytes, h._small_direct_bytes          # back to the class defaults
                else:
                    h._small_poll_bytes, h._small_direct_bytes = poll, direct
                p = pcts(lambda i: h.hash_batch_packed(xs[(i * n) % 3968:(i * n) % 3968 + n]), 300)
                row.append(f"{label}: {p[0]:6.1f} /{p[1]:6.1f}")
            print("   ".join(row), flush=True)
        h.close()


if __name__ == "__main__":
    main()
