#!/usr/bin/env python3
"""GPU box: rows whose split residual is aligned with one hyperplane (tests/_adversary.py) through hash_device in each
window mode; prints how many of the targeted bits differ from the reference-literal NumPy path."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lshrs_amd import LSHHasher  # noqa: E402
from oracle import lshrs_oracle as O  # noqa: E402
from tests._adversary import adversarial_row, describe  # noqa: E402


def main():
    dim = int(sys.argv[1]) if len(sys.argv) > 1 else 768
    modes = [("default (proven window)", {}), ("window64, no guard", {"tau1_ulps": 64.0 * (768.0 / dim) ** 0.5, "margin_guard": 0.0}),
             ("window64 + guard", {"tau1_ulps": 64.0 * (768.0 / dim) ** 0.5})]
    for name, kw in modes:
        h = LSHHasher(16, 16, dim, seed=42, **kw)
        rng = np.random.default_rng(3)
        x = rng.standard_normal((4096, dim)).astype(np.float32)
        targets = []
        for i, (band, bit) in enumerate([(b, r) for b in range(16) for r in (0, 5, 11, 15)]):
            for sign in (1.0, -1.0):
                row = 17 + 31 * len(targets)
                x[row] = adversarial_row(sign * h.projections[band][bit], 20.0, seed=i)
                targets.append((row, band, bit))
        want = O.hash_batch_literal_packed(h.projections, x)
        got = h.hash_device(torch.from_numpy(x).cuda()).cpu().numpy()
        bad_rows = np.flatnonzero((got != want).any(axis=(1, 2)))
        hit = sum(1 for (row, band, bit) in targets if (got[row, band, bit >> 3] ^ want[row, band, bit >> 3]) >> (bit & 7) & 1)
        d = describe(x[targets[0][0]], h.projections[targets[0][1]][targets[0][2]])
        print(f"{name}: dim {dim} window {h.tau1_ulps:.1f} mode {h.window_mode['tau1']}: {len(bad_rows)} rows differ, "
              f"{hit} of {len(targets)} targeted bits wrong; stats {h.last_stats}; first target {d}", flush=True)


if __name__ == "__main__":
    main()
