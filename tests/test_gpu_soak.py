"""Randomised end-to-end check of the default path (split-precision pass where it applies, pipelined tie-break, native
host engine) against the plain single-launch exact-f32 path, which the other GPU tests pin to the oracle: random
shapes, batch sizes from one row to 1.5 M, zero and NaN rows, row flags."""

from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def test_default_path_equals_plain_f32_path_on_random_batches(torch_mod):
    torch = torch_mod
    from lshrs_amd import LSHHasher

    rng = np.random.default_rng(77)
    shapes = [(16, 16, 768), (16, 32, 1536), (16, 16, 128), (32, 8, 96), (16, 16, 100), (5, 12, 64)]
    hashers = {}
    seen_split = seen_piped = seen_replay = 0
    for it in range(24):
        nb, r, dim = shapes[it % len(shapes)]
        n = int((int(rng.integers(1, 5000)), int(rng.integers(60_000, 300_000)), int(rng.integers(300_000, 1_500_000)))[it % 3])
        n = min(n, 900_000_000 // dim)
        if (nb, r, dim) not in hashers:
            # the two host-route hashers with round 2's measured windows (the mechanics under test: chunking, overlap, lists;
            # the proven windows on that route: tests/test_gpu_signature.py::test_host_engine_route_with_the_proven_windows);
            # `fast` is the default hasher: proven windows, ties replayed on the device
            measured = dict(tau_ulps=8.0, tau1_ulps=64.0)
            plain = LSHHasher(nb, r, dim, seed=11, precision="f32", tie_replay="off", **measured)
            plain.pipeline_chunk_rows = 10**9
            hashers[(nb, r, dim)] = (LSHHasher(nb, r, dim, seed=11), plain,
                                     LSHHasher(nb, r, dim, seed=11, tie_replay="off", **measured))
        fast, plain, fast_host = hashers[(nb, r, dim)]
        x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(1000 + it))
        if n > 10:
            x[int(rng.integers(0, n))] = 0.0
            x[int(rng.integers(0, n)), int(rng.integers(0, dim))] = float("nan")
        fa = torch.zeros(n, dtype=torch.uint8, device="cuda")
        fb = torch.zeros(n, dtype=torch.uint8, device="cuda")
        ka = fast.hash_device(x, row_flags=fa)
        seen_split += bool(fast._split_applies(n))
        seen_replay += fast.last_stats.get("tie_break_engine") == "device-replay"
        kb = plain.hash_device(x, row_flags=fb)
        assert torch.equal(ka, kb), f"keys differ: shape {(nb, r, dim)}, n = {n}"
        assert torch.equal(fa, fb), f"row flags differ: shape {(nb, r, dim)}, n = {n}"
        kf = LSHHasher(nb, r, dim, seed=11, precision="f32").hash_device(x) if it % 4 == 0 else kb   # f32 kernel + device replay
        assert torch.equal(kf, kb), f"f32 kernel + device tie replay: keys differ: shape {(nb, r, dim)}, n = {n}"
        kh = fast_host.hash_device(x)                       # ties broken on the host (chunked when the batch is large)
        seen_piped += "t_total_ms" in fast_host.last_stats
        assert torch.equal(kh, kb), f"host tie-break: keys differ: shape {(nb, r, dim)}, n = {n}"
        assert fast_host.last_stats["tie_pairs"] == plain.last_stats["tie_pairs"]
    assert seen_split >= 6 and seen_piped >= 6        # the batch mix really exercised both
    assert seen_replay == 0 or seen_replay >= seen_split   # (0: this host's BLAS order is not one the replay knows)
