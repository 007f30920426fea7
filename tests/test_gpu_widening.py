"""SURVEY §8f rows 3 and 4 THROUGH the HIP hasher: the Parquet fast path feeding
``LSHRS.create_signatures`` and indices restored from disk / from a pickle hashing on the device - against the oracle's
literal restatement of the reference (lshrs/io/parquet.py:206-227, lshrs/core/main.py:315-384, :846-1044)."""

from __future__ import annotations

import os
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


@pytest.mark.parametrize("n,dim,num_perm,kind,batch", [(6_000, 128, 64, "list_f32", 2_500), (33_000, 768, 256, "fixed_f32", 33_000),
                                                       (5_000, 96, 128, "list_f64", 10_000)])
def test_parquet_file_to_buckets_through_the_hip_hasher(torch_mod, tmp_path, n, dim, num_perm, kind, batch):
    """A Parquet file written on this box -> `create_signatures(format="parquet")` -> buckets.  The loader is this
    build's array-based one (parquet_fast), the hasher the HIP one (the 33 000 x 768 case streams through the split pass),
    the store array-fed; the expected buckets come from the literal loader + the literal per-vector loop."""
    pa = pytest.importorskip("pyarrow")
    import pyarrow.parquet as pq

    from lshrs_amd import LSHRS, InMemoryStorage
    from oracle.lshrs_oracle import index_literal
    from oracle.parquet_oracle import iter_parquet_vectors_literal

    rng = np.random.default_rng(n)
    data = rng.standard_normal((n, dim))
    ids = rng.permutation(10 ** 7)[:n].astype(np.int64)
    if kind == "list_f32":
        col = pa.array(data.astype(np.float32).tolist(), type=pa.list_(pa.float32()))
    elif kind == "list_f64":
        col = pa.array(data.tolist(), type=pa.list_(pa.float64()))
    else:
        col = pa.FixedSizeListArray.from_arrays(pa.array(data.astype(np.float32).reshape(-1)), dim)
    path = str(tmp_path / "vectors.parquet")
    pq.write_table(pa.table({"index": pa.array(ids), "vector": col}), path, row_group_size=7_000)

    for packed in (True, False):
        if not packed and n > 10_000:
            continue                                          # (op tuples: 16 Python objects per vector - the small cases)
        idx = LSHRS(dim=dim, num_perm=num_perm, storage=InMemoryStorage(), packed_ingest=packed, buffer_size=4_000)
        idx.create_signatures(format="parquet", source=path, batch_size=batch)
        idx.flush()
        want = InMemoryStorage()
        for bi, bv in iter_parquet_vectors_literal(path, batch_size=batch):
            index_literal(want, bi, bv, idx._hasher.projections, dim, 4_000)
        assert idx._storage.bucket_contents() == want.bucket_contents(), (kind, packed)
        if not packed:                                        # the reference's flush boundaries too (lshrs/core/main.py:1125-1143)
            assert [len(b) for b in idx._storage.batches] == [len(b) for b in want.batches]
    q = data[:3].astype(np.float32)
    assert [r[0] for r in idx.query_many(q, top_k=1)] == [int(i) for i in ids[:3]]


def test_restored_indices_hash_on_the_device(torch_mod, golden_dir, tmp_path):
    """f4: `load_from_disk` of the index the REFERENCE saved (tests/golden/g7_saved_index), a save / load round trip of a
    768-d index written here, and pickle round trips of both - every restored hasher re-uploads the stored hyperplanes
    and its device keys are the literal NumPy path's with those hyperplanes (lshrs/core/main.py:898-983, :1010-1044)."""
    torch = torch_mod
    from lshrs_amd import LSHRS, InMemoryStorage
    from oracle.lshrs_oracle import hash_batch_literal_packed

    ref = LSHRS.load_from_disk(os.path.join(golden_dir, "g7_saved_index"), storage=InMemoryStorage())
    big = LSHRS(dim=768, num_perm=256, storage=InMemoryStorage(), seed=77)
    # hyperplanes that are NOT the seed's draw: what comes back must be what was stored, not what the seed would give
    big._hasher.projections = [(p * np.float32(0.5) + np.float32(0.01)).astype(np.float32) for p in big._hasher.projections]
    big.save_to_disk(tmp_path / "big")
    loaded = LSHRS.load_from_disk(tmp_path / "big", storage=InMemoryStorage())
    assert all(np.array_equal(a, b) for a, b in zip(loaded._hasher.projections, big._hasher.projections))
    for idx, n in ((ref, 3_000), (loaded, 40_000)):
        dim = idx._config["dim"]
        x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(n))
        sl = slice(n - 2_000, n)
        want = hash_batch_literal_packed(idx._hasher.projections, x[sl].cpu().numpy())
        keys = idx._hasher.hash_device(x)
        assert np.array_equal(keys[sl].cpu().numpy(), want)
        clone = pickle.loads(pickle.dumps(idx))
        clone._storage = InMemoryStorage()                    # (the pickle carries no storage: lshrs/core/main.py:1010-1044)
        assert torch.equal(clone._hasher.hash_device(x), keys)
        assert clone._hasher.hash_vector(x[5].cpu().numpy()).as_tuple() == tuple(
            bytes(k) for k in hash_batch_literal_packed(idx._hasher.projections, x[5:6].cpu().numpy())[0])
        # the restored index works end to end: ingest through the device, query finds the vector
        host = x[:500].cpu().numpy()
        clone.index(list(range(500)), host)
        assert clone.get_top_k(host[7], topk=1) == [7]


def test_in_process_multi_device_ingestion(torch_mod):
    """SURVEY §8(e) behind the API: `devices=[...]` cuts every large host batch into one contiguous row
    slice per entry, hashes the slices concurrently (one thread + hasher + streams per entry) and returns the keys in the
    original row order - so `LSHRS.index` enqueues exactly the operations a single device produces, in the same order.
    On the one-GPU box: devices=[0, 0] (two slices in flight on one device) against the single-device result, the oracle,
    and the reference's op order."""
    torch = torch_mod
    from lshrs_amd import LSHRS, InMemoryStorage, LSHHasher
    from oracle.lshrs_oracle import hash_batch_literal_packed

    n, dim = 150_000, 768
    x = np.random.default_rng(8).standard_normal((n, dim)).astype(np.float32)
    x[77_777] = 0.0                                                    # a zero vector (row flag bit 0) in the second slice
    one = LSHHasher(16, 16, dim, seed=42)
    two = LSHHasher(16, 16, dim, seed=42, devices=[0, 0])
    k1, f1 = one.hash_batch_packed(x, return_row_flags=True)
    k2, f2 = two.hash_batch_packed(x, return_row_flags=True)
    st = dict(two.last_stats)
    assert np.array_equal(k1, k2) and np.array_equal(f1, f2) and f2[77_777] & 1
    assert st["devices"] == [0, 0] and len(st["per_device"]) == 2 and sum(p["n"] for p in st["per_device"]) == n
    assert all(p.get("tie_break_engine") == "device-replay" for p in st["per_device"])
    sl = slice(74_000, 77_000)                                         # straddles the slice boundary (75 008)
    assert np.array_equal(k2[sl], hash_batch_literal_packed(two.projections, x[sl]))
    # small batches and single vectors stay on devices[0]; device tensors are hashed where they live
    assert "devices" not in (two.hash_batch_packed(x[:1000]), two.last_stats)[1]
    assert two.hash_vector(x[5]).as_tuple() == one.hash_vector(x[5]).as_tuple()
    # re-assigned hyperplanes reach every slice's hasher
    planes = [(-p).copy() for p in two.projections]
    two.projections = planes
    one.projections = planes
    assert np.array_equal(two.hash_batch_packed(x), one.hash_batch_packed(x))
    two.close()

    # four entries: a ragged batch - slices of whole 256-row tiles, the last one short -, fewer rows than
    # entries x the minimum (stays on devices[0]), and a worker that raises: the error reaches the caller, the hasher stays usable
    four = LSHHasher(16, 16, dim, seed=42, devices=[0, 0, 0, 0])
    four.multi_device_min_rows = 1_000
    ragged = x[:4 * 1_000 + 777]
    k4 = four.hash_batch_packed(ragged)
    assert four.last_stats.get("devices") == [0, 0, 0, 0] and len(four.last_stats["per_device"]) == 4
    assert [st["n"] for st in four.last_stats["per_device"]] == [1280, 1280, 1280, 937]
    ref4 = LSHHasher(16, 16, dim, seed=42)
    assert np.array_equal(k4, ref4.hash_batch_packed(ragged))
    four.hash_batch_packed(x[:3_999])
    assert "devices" not in four.last_stats                            # 3 999 < 4 x 1 000: one device
    boom = four._children[2].hash_batch_packed

    def failing(*a, **kw):
        raise RuntimeError("worker 2 fell over")

    four._children[2].hash_batch_packed = failing
    with pytest.raises(RuntimeError, match="worker 2 fell over"):
        four.hash_batch_packed(ragged)
    four._children[2].hash_batch_packed = boom
    assert np.array_equal(four.hash_batch_packed(ragged), k4)          # (nothing is left half-done: the lock is free, the pool alive)
    four.close()

    # through the orchestrator: the reference's operation order and flush boundaries (lshrs/core/main.py:1125-1143)
    m = 3_000
    a = LSHRS(dim=dim, num_perm=256, storage=InMemoryStorage(), buffer_size=10_000, packed_ingest=False)
    b = LSHRS(dim=dim, num_perm=256, storage=InMemoryStorage(), buffer_size=10_000, devices=[0, 0], packed_ingest=False)
    b._hasher.multi_device_min_rows = 512
    ids = list(range(100, 100 + m))
    a.index(ids, x[:m])
    b.index(ids, x[:m])
    assert b._hasher.last_stats.get("devices") == [0, 0]
    assert a._storage.batches == b._storage.batches and len(a._storage.batches) == 5      # 48 000 ops, 10 000 at a time
    # ... and with array-fed buckets
    c = LSHRS(dim=dim, num_perm=256, storage=InMemoryStorage(), packed_ingest=True, devices=[0, 0])
    with pytest.raises(ValueError, match="zero"):
        c.index(np.arange(n), x)
    d = LSHRS(dim=dim, num_perm=256, storage=InMemoryStorage(), packed_ingest=True)
    with pytest.raises(ValueError, match="zero"):
        d.index(np.arange(n), x)
    # (both stop at the zero vector, having stored the rows in front of it - the reference's error timing)
    assert c._storage.bucket_contents() == d._storage.bucket_contents()
