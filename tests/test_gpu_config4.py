"""BASELINE config 4 (10M x 768-d, num_perm 256, ingest sharded across 8 GPUs) on the ONE device the GPU suite has:
every rank's 1.25M-row shard with that rank's rows (`torch.Generator(device).manual_seed(1000 + rank)`),
50 000 rows of each bit-compared with the oracle, and the whole 10M x 768 batch (30.7 GB, one allocation) through the
size-independent properties of tests/test_gpu_signature.py::test_full_size_properties_1m_rows.  The path has no collective:
a shard hashed alone on one device IS what its rank computes on an 8-GPU node (tests/test_sharding_gloo.py covers the
rendezvous / partitioning logic with two gloo ranks on the CPU)."""

from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHARD = 1_250_000


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def _check(keys_slice, h, x_slice):
    from oracle.parallel import SharedVectors, hash_shared_literal_packed

    m = int(x_slice.shape[0])
    with SharedVectors(m, 768) as sv:
        sv.array[:] = x_slice.cpu().numpy()
        want = hash_shared_literal_packed(h.projections, sv)
    return int((keys_slice.cpu().numpy() != want).any(axis=(1, 2)).sum())


def test_every_ranks_shard_of_config4(torch_mod):
    """The eight 1.25M x 768 shards bench.py --gpus 8 hashes (same generator seeds, same hasher), one after the other on
    this device: no list overflow, the proven window, and the first 50 000 rows of each shard byte-identical to the
    reference-literal loop ("CPU bit-check on the first 50 k rows of each shard")."""
    torch = torch_mod
    from lshrs_amd import LSHHasher
    from lshrs_amd.sharding import shard_range

    h = LSHHasher(16, 16, 768, seed=42)
    dev = torch.device("cuda", torch.cuda.current_device())
    keys = torch.empty((SHARD, 16, 2), dtype=torch.uint8, device=dev)
    for rank in range(8):
        assert shard_range(10_000_000, 8, rank) == (rank * SHARD, (rank + 1) * SHARD)
        x = torch.randn(SHARD, 768, device=dev, generator=torch.Generator(device=dev).manual_seed(1000 + rank))
        h.hash_device(x, out=keys)
        st = dict(h.last_stats)
        assert st["relaunches"] == 0 and st["tie_break_engine"] == "device-replay" and st["window"] == "proven", (rank, st)
        assert st["max_dev_units"] * 4 <= st["tau1_ulps"], (rank, st)
        assert _check(keys[:50_000], h, x[:50_000]) == 0, rank
        del x


def test_config4_all_10m_rows_on_one_device(torch_mod):
    """10M x 768 = 30.7 GB resident (config 4's whole workload on one of the 288 GB devices): one launch over all of it.
    Size-independent properties: the shards hashed alone give the bytes they have inside the big batch (= what the ranks
    of an 8-GPU run produce), hashing again gives the same bytes, power-of-two scaling changes nothing, negation flips
    every bit of a non-zero projection, half the bits are set; and 50 000 rows in the middle against the oracle."""
    torch = torch_mod
    from lshrs_amd import LSHHasher

    h = LSHHasher(16, 16, 768, seed=42)
    dev = torch.device("cuda", torch.cuda.current_device())
    n = 8 * SHARD
    x = torch.empty((n, 768), dtype=torch.float32, device=dev)
    for rank in range(8):
        x[rank * SHARD:(rank + 1) * SHARD].normal_(generator=torch.Generator(device=dev).manual_seed(1000 + rank))
    keys = h.hash_device(x)
    st = dict(h.last_stats)
    assert keys.shape == (n, 16, 2) and st["relaunches"] == 0 and st["window"] == "proven", st
    assert st["flagged"] > 1_000_000 and st["max_dev_units"] * 4 <= st["tau1_ulps"], st
    assert torch.equal(h.hash_device(x), keys)                                             # idempotent
    for rank in (0, 3, 7):                                                                 # a rank's shard, hashed alone
        lo = rank * SHARD
        assert torch.equal(h.hash_device(x[lo:lo + SHARD]), keys[lo:lo + SHARD]), rank
    x.mul_(4.0)                                                                            # exact: every sign unchanged
    assert torch.equal(h.hash_device(x), keys)
    x.mul_(-0.25)
    kneg = h.hash_device(x)
    assert (kneg ^ keys).eq(0xFF).float().mean().item() > 0.999999
    x.neg_()
    ones = torch.tensor([bin(i).count("1") for i in range(256)], device=dev)[keys.long()].sum().item()
    assert abs(ones / (n * 256) - 0.5) < 1e-3
    assert _check(keys[5_000_000:5_050_000], h, x[5_000_000:5_050_000]) == 0
