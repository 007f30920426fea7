"""The N>1 path on CPU: world_size-2 ``gloo`` processes, oracle-backed hasher double.
Row-sharded hashing/indexing must reproduce the single-process result exactly."""

from __future__ import annotations

import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from lshrs_amd.sharding import shard_range

    for n in (0, 1, 7, 8, 9, 1_000_000, 10_000_000):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(10_000_000, 8, 3) == (3_750_000, 5_000_000)
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist

    from lshrs_amd import LSHRS, InMemoryStorage
    from lshrs_amd.sharding import hash_sharded, index_sharded, world_info
    from tests._doubles import OracleBackedHasher

    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        assert world_info() == (rank, world, rank)
        x = np.random.default_rng(99).standard_normal((1001, 48)).astype(np.float32)   # odd n: ragged shards
        hasher = OracleBackedHasher(6, 8, 48, seed=5)                                 # replicated hyperplanes
        full = hash_sharded(hasher, x, gather=True)
        local = hash_sharded(hasher, x)
        store = InMemoryStorage()
        idx = LSHRS(dim=48, num_bands=6, rows_per_band=8, num_perm=48, storage=store, hasher=hasher, buffer_size=500)
        lo, hi = index_sharded(idx, list(range(1001)), x)
        ops = [op for batch in store.batches for op in batch]
        gathered = [None] * world
        dist.all_gather_object(gathered, (lo, hi, ops))
        dist.barrier()
        if rank == 0:
            np.save(os.path.join(out_dir, "full.npy"), full)
            np.save(os.path.join(out_dir, "local0.npy"), local)
            import pickle

            pickle.dump(gathered, open(os.path.join(out_dir, "ops.pkl"), "wb"))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_equals_single_process(tmp_path):
    import pickle

    import torch.multiprocessing as mp

    from lshrs_amd import LSHRS, InMemoryStorage
    from tests._doubles import OracleBackedHasher

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)

    x = np.random.default_rng(99).standard_normal((1001, 48)).astype(np.float32)
    hasher = OracleBackedHasher(6, 8, 48, seed=5)
    want = hasher.hash_batch_packed(x)
    assert np.array_equal(np.load(tmp_path / "full.npy"), want)
    assert np.array_equal(np.load(tmp_path / "local0.npy"), want[:501])
    gathered = pickle.load(open(tmp_path / "ops.pkl", "rb"))
    assert [(lo, hi) for lo, hi, _ in gathered] == [(0, 501), (501, 1001)]
    single = InMemoryStorage()
    LSHRS(dim=48, num_bands=6, rows_per_band=8, num_perm=48, storage=single, hasher=hasher,
          buffer_size=10**9).index(list(range(1001)), x)
    single_ops = [op for batch in single.batches for op in batch]
    sharded_ops = [op for _, _, ops in gathered for op in ops]
    assert sharded_ops == single_ops          # rank-major concatenation == original row order


def _engine_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    """Both ranks of a node build their host tie-break engine AT THE SAME TIME (LOCAL_WORLD_SIZE = 2 halves each one's
    worker budget) and patch the same (row, band) pairs while the other does."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    os.environ.pop("LSHRS_TIE_THREADS", None)
    import pickle

    import torch.distributed as dist

    from lshrs_amd import _hostblas

    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(7)
        planes = rng.standard_normal((4, 16, 256)).astype(np.float32)
        xs = rng.standard_normal((3_000, 256)).astype(np.float32)
        rows = rng.integers(0, 3_000, 20_000).astype(np.int32)
        bands = np.sort(rng.integers(0, 4, 20_000)).astype(np.int32)
        dist.barrier()
        eng = _hostblas.engine()
        info = {"budget": _hostblas._core_budget(), "default_threads": _hostblas.default_threads(),
                "threads": None if eng is None else eng.threads}
        keys = None
        if eng is not None:
            assert eng.shape_trusted(planes)
            dist.barrier()                                       # both engines exist: now both work at once
            keys = eng.patch(planes, xs, rows, bands)
        want = np.stack([np.packbits((planes[b] @ xs[r]) > 0, bitorder="little") for r, b in zip(rows, bands)])
        info["equal_numpy"] = None if keys is None else bool(np.array_equal(keys, want))
        gathered = [None] * world
        dist.all_gather_object(gathered, (info, None if keys is None else keys.tobytes()))
        if rank == 0:
            pickle.dump(gathered, open(os.path.join(out_dir, "engines.pkl"), "wb"))
    finally:
        dist.destroy_process_group()


def test_two_ranks_run_their_host_engines_side_by_side(tmp_path):
    """a deployment whose BLAS the replay does not know runs the host engine on every rank; the ranks
    of a node divide the core budget (`_hostblas.default_threads`, LOCAL_WORLD_SIZE) - here two gloo ranks build their
    engines under that division and patch the same pairs concurrently: same bytes on both, equal to NumPy's
    `P_band @ x` (the reference's call, lshrs/hash/lsh.py:200)."""
    import pickle

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_engine_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    (i0, k0), (i1, k1) = pickle.load(open(tmp_path / "engines.pkl", "rb"))
    assert i0["budget"] == i1["budget"] and i0["default_threads"] == i1["default_threads"]
    assert i0["default_threads"] == max(1, min(8, i0["budget"] // 2 // 2))          # halved by LOCAL_WORLD_SIZE = 2
    if i0["threads"] is None:
        pytest.skip(f"no engine on this host (budget {i0['budget']} cores over 2 ranks: NumPy's own call is used)")
    assert i0["threads"] == i1["threads"] == i0["default_threads"]
    assert i0["equal_numpy"] and i1["equal_numpy"] and k0 == k1
