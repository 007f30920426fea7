"""Host logic of ``LSHRS`` (buffering, flush boundaries, error timing, query ordering, validation)
against the reference's own recorded behaviour (tests/golden/g5_orchestration.json) and the
behavioural assertions of the reference suite (tests/test_core.py, test_buffer_semantics.py,
test_concurrency.py upstream).  CPU only: hashing/reranking are supplied by the oracle double
(tests/_doubles.py); tests/test_gpu_orchestrator.py repeats the golden checks through the HIP path.
"""

from __future__ import annotations

import hashlib
import json
import os
import threading

import numpy as np
import pytest

from lshrs_amd import LSHRS, InMemoryStorage
from tests._doubles import make_cpu_lshrs


@pytest.fixture(scope="module")
def g5(golden_dir):
    return json.load(open(os.path.join(golden_dir, "g5_orchestration.json")))


def ops_json(batch):
    return [[int(b), k.hex(), int(i)] for b, k, i in batch]


def small(monkeypatch, **kw):
    kw.setdefault("dim", 32)
    kw.setdefault("num_bands", 4)
    kw.setdefault("rows_per_band", 4)
    kw.setdefault("num_perm", 16)
    return make_cpu_lshrs(monkeypatch, **kw)


# ----------------------------------------------------------------------------- goldens
def test_index_batches_equal_reference(monkeypatch, g5):
    data = np.random.default_rng(301).standard_normal((50, 32)).astype(np.float32)
    store = InMemoryStorage()
    idx = small(monkeypatch, buffer_size=10, storage=store, vector_fetch_fn=lambda ids: data[np.asarray(ids)])
    idx.index(list(range(50)), data)
    assert [ops_json(b) for b in store.batches] == g5["index50"]["batches"]
    queries = np.frombuffer(bytes.fromhex(g5["queries_hex"]), dtype=np.float32).reshape(5, 32)
    assert [idx.get_top_k(q, topk=5) for q in queries] == g5["top_k_5"]
    for q, want in zip(queries, g5["above_p_half"]):
        got = idx.get_above_p(q, p=0.5)
        assert [i for i, _ in got] == [i for i, _ in want]
        assert np.allclose([s for _, s in got], [s for _, s in want], atol=1e-6)
        assert all(isinstance(i, int) and isinstance(s, float) for i, s in got)
    for q, want in zip(queries, g5["query_topk3_topp1"]):
        got = idx.query(q, top_k=3, top_p=1.0)
        assert [i for i, _ in got] == [i for i, _ in want]


def test_zero_vector_mid_batch_error_timing(monkeypatch, g5):
    bad = np.frombuffer(bytes.fromhex(g5["bad_hex"]), dtype=np.float32).reshape(12, 32)
    store = InMemoryStorage()
    idx = small(monkeypatch, buffer_size=10, storage=store)
    with pytest.raises(ValueError) as exc:
        idx.index(list(range(12)), bad)
    want = g5["zero_row7"]
    assert str(exc.value) == want["message"]
    assert [ops_json(b) for b in store.batches] == want["batches"]
    assert ops_json(idx._buffer) == want["left_in_buffer"]


def test_negative_id_mid_batch_error_timing(monkeypatch, g5):
    bad = np.frombuffer(bytes.fromhex(g5["bad_hex"]), dtype=np.float32).reshape(12, 32)
    store = InMemoryStorage()
    idx = small(monkeypatch, buffer_size=1000, storage=store)
    with pytest.raises(ValueError) as exc:
        idx.index([0, 1, 2, -4, 5], bad[:5])
    want = g5["negative_row3"]
    assert str(exc.value) == want["message"]
    assert [ops_json(b) for b in store.batches] == want["batches"] == []
    assert ops_json(idx._buffer) == want["left_in_buffer"]


def test_config1_plumbing_10k_x_128(monkeypatch, g5):
    """BASELINE config 1: 10k x 128-d, num_perm=64 -> (16, 4), in-memory storage, CPU plumbing."""
    store = InMemoryStorage()
    # (packed_ingest=False: the reference's operation lists and flush boundaries are what g5 pins; the default, "auto",
    #  hands a batch of this size over as ONE bucket CSR - same buckets, see test_packed_ops.py)
    idx = make_cpu_lshrs(monkeypatch, dim=128, num_perm=64, storage=store, buffer_size=10_000, packed_ingest=False)
    x = np.random.default_rng(1).standard_normal((10_000, 128)).astype(np.float32)
    idx.index(list(range(10_000)), x)
    want = g5["c1"]
    assert (idx._hasher.num_bands, idx._hasher.rows_per_band) == (want["num_bands"], want["rows_per_band"])
    assert [len(b) for b in store.batches] == want["batches"]
    h = hashlib.sha256()
    for batch in store.batches:
        for b, k, i in batch:
            h.update(bytes([b]) + k + int(i).to_bytes(4, "little"))
    assert h.hexdigest() == want["ops_sha256"]
    assert idx.get_top_k(x[0], topk=5) == want["top_k_row0"]


# ----------------------------------------------------------------------------- constructor / validation
def test_constructor_validation():
    s = InMemoryStorage()
    with pytest.raises(ValueError, match="dimensionality"):
        LSHRS(dim=0, storage=s, hasher=object())
    with pytest.raises(ValueError, match="num_perm"):
        LSHRS(dim=4, num_perm=0, storage=s, hasher=object())
    with pytest.raises(ValueError, match="buffer_size"):
        LSHRS(dim=4, buffer_size=0, storage=s, hasher=object())
    with pytest.raises(ValueError, match="must equal num_perm"):
        LSHRS(dim=4, num_perm=16, num_bands=3, rows_per_band=4, storage=s, hasher=object())
    idx = LSHRS(dim=8, num_perm=256, storage=s, hasher=object())
    assert (idx._config["num_bands"], idx._config["rows_per_band"]) == (16, 16)
    assert idx.stats()["num_perm"] == 256


def test_ingest_and_index_validation(monkeypatch):
    idx = small(monkeypatch)
    v = np.ones(32, dtype=np.float32)
    with pytest.raises(ValueError, match="non-negative"):
        idx.ingest(-1, v)
    with pytest.raises(ValueError, match="dimension"):
        idx.ingest(0, np.ones(31, dtype=np.float32))
    with pytest.raises(ValueError, match="zero vector"):
        idx.ingest(0, np.zeros(32, dtype=np.float32))
    idx.index([], None)  # no-op
    with pytest.raises(ValueError, match="shape"):
        idx.index([0, 1], np.ones((2, 31), dtype=np.float32))
    with pytest.raises(ValueError, match="does not match"):
        idx.index([0, 1, 2], np.ones((2, 32), dtype=np.float32))
    with pytest.raises(RuntimeError, match="vector_fetch_fn"):
        idx.index([0, 1], None)


def test_index_with_fetch_fn_and_self_match(monkeypatch):
    data = np.random.default_rng(3).standard_normal((40, 32)).astype(np.float32)
    idx = small(monkeypatch, vector_fetch_fn=lambda ids: data[np.asarray(ids)])
    idx.index(list(range(40)))
    assert idx._storage.total_operations == 40 * 4
    for i in (0, 17, 39):
        assert idx.get_top_k(data[i], topk=1) == [i]
        near = data[i] + 1e-4 * np.random.default_rng(i).standard_normal(32).astype(np.float32)
        assert i in idx.get_top_k(near, topk=3)
    res = idx.get_above_p(data[5], p=1.0)
    assert res[0][0] == 5 and res[0][1] == pytest.approx(1.0, abs=1e-5)
    assert [s for _, s in res] == sorted((s for _, s in res), reverse=True)


def test_query_validation(monkeypatch):
    data = np.random.default_rng(4).standard_normal((10, 32)).astype(np.float32)
    idx = small(monkeypatch)
    idx.index(list(range(10)), data)
    with pytest.raises(ValueError, match="top_k"):
        idx.query(data[0], top_k=0)
    with pytest.raises(ValueError, match="top_p"):
        idx.query(data[0], top_k=None, top_p=1.5)
    with pytest.raises(ValueError, match="top_p"):
        idx.query(data[0], top_k=None, top_p=0.0)
    with pytest.raises(RuntimeError, match="vector_fetch_fn"):
        idx.get_above_p(data[0], p=0.5)
    with pytest.raises(ValueError, match="zero vector"):
        idx.get_top_k(np.zeros(32, dtype=np.float32))
    with pytest.raises(ValueError, match="dimension"):
        idx.get_top_k(np.ones(3, dtype=np.float32))
    empty = small(monkeypatch, storage=InMemoryStorage())
    assert empty.get_top_k(data[0]) == []
    bad_fetch = small(monkeypatch, storage=idx._storage, vector_fetch_fn=lambda ids: np.ones((1, 32), np.float32))
    with pytest.raises(ValueError, match="mismatched batch size"):
        bad_fetch.get_above_p(data[0], p=1.0)
    bad_shape = small(monkeypatch, storage=idx._storage, vector_fetch_fn=lambda ids: np.ones((len(ids), 3), np.float32))
    with pytest.raises(ValueError, match="Fetched vectors must have shape"):
        bad_shape.get_above_p(data[0], p=1.0)


# ----------------------------------------------------------------------------- buffer semantics
def test_buffered_until_flush_and_close(monkeypatch):
    store = InMemoryStorage()
    idx = small(monkeypatch, storage=store, buffer_size=1000)
    v = np.random.default_rng(0).standard_normal(32).astype(np.float32)
    idx.ingest(1, v)
    assert store.batches == [] and len(idx._buffer) == 4
    assert idx.get_top_k(v) == []          # not visible before the flush
    idx.flush()
    assert len(store.batches) == 1 and idx._buffer == []
    assert idx.get_top_k(v) == [1]
    idx.flush()                            # empty flush is a no-op
    assert len(store.batches) == 1
    idx.ingest(2, v)
    idx.close()
    assert store.closed and len(store.batches) == 2


def test_auto_flush_boundary(monkeypatch):
    store = InMemoryStorage()
    idx = make_cpu_lshrs(monkeypatch, dim=32, num_bands=2, rows_per_band=4, num_perm=8, buffer_size=4, storage=store)
    rng = np.random.default_rng(1)
    idx.ingest(0, rng.standard_normal(32).astype(np.float32))
    assert store.batches == []
    idx.ingest(1, rng.standard_normal(32).astype(np.float32))
    assert [len(b) for b in store.batches] == [4]


def test_flush_failure_restores_buffer(monkeypatch):
    store = InMemoryStorage(fail_on_flush=True)
    idx = small(monkeypatch, storage=store, buffer_size=1000)
    v = np.random.default_rng(2).standard_normal(32).astype(np.float32)
    idx.ingest(7, v)
    before = list(idx._buffer)
    with pytest.raises(ConnectionError):
        idx.flush()
    assert idx._buffer == before
    store._fail_on_flush = False
    idx.ingest(8, v)
    idx.flush()
    assert [i for _, _, i in store.batches[0]] == [7] * 4 + [8] * 4   # restored ops stay in front


def test_context_manager_delete_clear(monkeypatch):
    store = InMemoryStorage()
    data = np.random.default_rng(5).standard_normal((6, 32)).astype(np.float32)
    with small(monkeypatch, storage=store) as idx:
        idx.index(list(range(6)), data)
        idx.delete(3)
        assert 3 not in idx.get_top_k(data[3], topk=6)
        idx.delete([0, 1])
        assert idx.get_top_k(data[0], topk=6) == [] or 0 not in idx.get_top_k(data[0], topk=6)
        idx.clear()
        assert idx.get_top_k(data[4]) == []
    assert store.closed


# ----------------------------------------------------------------------------- concurrency
def test_concurrent_ingest_and_flush(monkeypatch):
    store = InMemoryStorage()
    idx = small(monkeypatch, storage=store, buffer_size=37)
    data = np.random.default_rng(6).standard_normal((100, 32)).astype(np.float32)
    errors = []

    def worker(t):
        try:
            for j in range(10):
                idx.ingest(t * 10 + j, data[t * 10 + j])
        except Exception as exc:  # pragma: no cover
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(10)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    flushers = [threading.Thread(target=idx.flush) for _ in range(5)]
    [t.start() for t in flushers]
    [t.join() for t in flushers]
    assert not errors
    assert store.total_operations == 100 * 4
    assert store.unique_indices == set(range(100))


def test_same_seed_same_keys_different_seed_differs(monkeypatch):
    data = np.random.default_rng(8).standard_normal((20, 32)).astype(np.float32)
    a, b, c = InMemoryStorage(), InMemoryStorage(), InMemoryStorage()
    small(monkeypatch, storage=a, seed=1).index(list(range(20)), data)
    small(monkeypatch, storage=b, seed=1).index(list(range(20)), data)
    small(monkeypatch, storage=c, seed=2).index(list(range(20)), data)
    assert a.batches == b.batches
    assert a.batches != c.batches


def test_create_signatures_from_batches(monkeypatch):
    data = np.random.default_rng(9).standard_normal((30, 32)).astype(np.float32)
    store = InMemoryStorage()
    idx = small(monkeypatch, storage=store)
    idx.create_signatures("batches", batches=[(list(range(0, 10)), data[:10]), (list(range(10, 30)), data[10:])])
    assert store.total_operations == 30 * 4 and len(store.batches) == 2
    with pytest.raises(ValueError, match="Unsupported"):
        idx.create_signatures("csv")


def test_query_many_equals_looping_query(monkeypatch):
    """Batched multi-query == [query(v) for v in vectors], ragged candidate lists included."""
    rng = np.random.default_rng(77)
    data = rng.standard_normal((300, 32)).astype(np.float32)
    idx = small(monkeypatch, vector_fetch_fn=lambda ids: data[np.asarray(ids)], buffer_size=100_000)
    idx.index(list(range(300)), data)
    queries = np.concatenate([data[:20] + 0.05 * rng.standard_normal((20, 32)).astype(np.float32),
                              rng.standard_normal((5, 32)).astype(np.float32) * 50.0])
    for kw in ({"top_k": 5, "top_p": None}, {"top_k": None, "top_p": None}, {"top_k": None, "top_p": 0.5},
               {"top_k": 3, "top_p": 1.0}, {"top_k": None, "top_p": 1e-9}):
        many = idx.query_many(queries, **kw)
        single = [idx.query(q, **kw) for q in queries]
        assert len(many) == len(single) == 25
        for a, b in zip(many, single):
            if kw["top_p"] is None:
                assert a == b
            else:
                assert [i for i, _ in a] == [i for i, _ in b]
                assert np.allclose([s for _, s in a], [s for _, s in b], atol=1e-6)
    assert idx.query_many(np.empty((0, 32), dtype=np.float32)) == []
    # the array form (round 6): ids / scores / bounds carry exactly the lists; engines other than the host's need the HIP hasher
    for kw in ({"top_k": 5, "top_p": None}, {"top_k": None, "top_p": 0.5}, {"top_k": 2, "top_p": 1.0}):
        lists = idx.query_many(queries, **kw)
        ids, scores, bounds = idx.query_many(queries, return_arrays=True, **kw)
        assert ids.dtype == np.int64 and bounds.dtype == np.int64 and bounds.shape == (26,) and bounds[0] == 0 and bounds[-1] == len(ids)
        if kw["top_p"] is None:
            assert scores is None and [ids[bounds[i]:bounds[i + 1]].tolist() for i in range(25)] == lists
        else:
            assert scores.dtype == np.float32 and len(scores) == len(ids)
            assert [list(zip(ids[bounds[i]:bounds[i + 1]].tolist(), scores[bounds[i]:bounds[i + 1]].astype(np.float64).tolist()))
                    for i in range(25)] == lists
    e_ids, e_scores, e_bounds = idx.query_many(np.empty((0, 32), dtype=np.float32), top_p=0.5, return_arrays=True)
    assert len(e_ids) == 0 and len(e_scores) == 0 and e_bounds.tolist() == [0]
    with pytest.raises(RuntimeError, match="HIP hasher"):
        idx.query_many(queries, engine="device")
    with pytest.raises(ValueError, match="engine must be"):
        idx.query_many(queries, engine="gpu")
    idx.set_corpus(data)                                     # (a host array: the host engine's rerank takes it like a tensor)
    with pytest.raises(ValueError, match="corpus must have shape"):
        idx.set_corpus(np.zeros((5, 31), np.float32))
    idx.set_corpus(None)
    with pytest.raises(ValueError, match="zero vector"):
        idx.query_many(np.zeros((2, 32), dtype=np.float32))
    with pytest.raises(ValueError, match="shape"):
        idx.query_many(np.ones((2, 31), dtype=np.float32))
    with pytest.raises(ValueError, match="top_k"):
        idx.query_many(queries, top_k=0)
    with pytest.raises(ValueError, match="top_p"):
        idx.query_many(queries, top_k=None, top_p=2.0)
    empty = small(monkeypatch, storage=InMemoryStorage(), vector_fetch_fn=lambda ids: data[np.asarray(ids)])
    assert empty.query_many(queries[:3], top_k=None, top_p=0.5) == [[], [], []]


@pytest.mark.parametrize("id_kind", ["composite-key sorts", "wide ids: lexsort"])
@pytest.mark.parametrize("packed", [False, True])
def test_array_collision_counting_equals_the_per_member_loop(monkeypatch, packed, id_kind):
    """`_ordered_candidates_many` (flat (query, member) pairs, one sort to count, one to order) against the reference's
    per-member dictionary loop + sort (lshrs/core/main.py:1101-1109, :614), on stores built from op tuples and from a
    bucket CSR, with crowded buckets, ties in the counts and arbitrary 64-bit ids."""
    rng = np.random.default_rng(0)
    idx = make_cpu_lshrs(monkeypatch, dim=16, num_bands=6, rows_per_band=3, num_perm=18, packed_ingest=packed)
    data = rng.standard_normal((3000, 16)).astype(np.float32)
    ids = rng.permutation(10**6)[:3000].astype(np.int64) * 4_000_003
    if id_kind.startswith("wide"):
        ids = ids * 1_000_003                                # past the bits a (query, count, id) key leaves for the id
    ids = ids.tolist()
    idx.index(ids[:2000], data[:2000])
    idx.index(ids[2000:], data[2000:])                       # two CSR segments / more batches
    queries = rng.standard_normal((150, 16)).astype(np.float32)
    keys, _ = idx._hasher.hash_batch_packed(queries, return_row_flags=True)
    many = idx._ordered_candidates_many(keys)
    for i in range(len(queries)):
        counts = idx._candidate_counts_from_keys(keys[i])
        assert many[i] == [k for k, _ in sorted(counts.items(), key=lambda it: (-it[1], it[0]))]
    assert max(len(m) for m in many) > 50


@pytest.mark.parametrize("nb,r,buffer_size,prefill", [(4, 4, 1, 0), (4, 4, 7, 2), (16, 4, 100, 3), (3, 5, 16, 1), (8, 2, 10_000, 5),
                                                       (16, 16, 33, 0), (5, 9, 4, 1)])
def test_op_tuple_windows_are_the_per_vector_loops_batches(monkeypatch, nb, r, buffer_size, prefill):
    """`index()` builds the operation tuples a flush WINDOW at a time (zipped, one lock acquisition) - the
    batches the store receives are exactly those of the reference's per-vector loop (oracle.index_literal: one vector, `num_bands`
    tuples, flush at the first vector boundary with >= buffer_size operations; lshrs/core/main.py:1113-1143), also with
    operations of earlier `ingest()` calls already in the buffer, a bad row in the middle, and ids that are big Python ints."""
    from oracle import lshrs_oracle as O

    dim = 24
    rng = np.random.default_rng(nb * 100 + buffer_size)
    data = rng.standard_normal((257, dim)).astype(np.float32)
    ids = [int(v) for v in rng.choice(10**6, 257, replace=False)]
    ids[5] = 2**70 + 3                                                   # (no int64: the tuples carry the caller's ints)
    got_store, want_store = InMemoryStorage(), InMemoryStorage()
    idx = make_cpu_lshrs(monkeypatch, dim=dim, num_bands=nb, rows_per_band=r, num_perm=nb * r, buffer_size=buffer_size,
                         storage=got_store, packed_ingest=False)
    P = idx._hasher.projections
    pre = rng.standard_normal((prefill, dim)).astype(np.float32)
    for j in range(prefill):
        idx.ingest(900_000 + j, pre[j])
    idx.index(ids, data)
    # the literal loop over the same sequence (the prefilled vectors first: one buffer)
    O.index_literal(want_store, [900_000 + j for j in range(prefill)] + ids, np.concatenate([pre, data]), P, dim, buffer_size)
    assert got_store.batches == want_store.batches
    assert all(type(b) is int and type(k) is bytes and type(i) is int for batch in got_store.batches for b, k, i in batch)
    assert got_store.bucket_contents() == want_store.bucket_contents()
    # a zero vector at row 100: everything in front of it is enqueued (and flushed where the loop flushes), then the error;
    # what is left in the buffer is what the reference leaves there
    bad = data.copy()
    bad[100] = 0
    s2, w2 = InMemoryStorage(), InMemoryStorage()
    idx2 = make_cpu_lshrs(monkeypatch, dim=dim, num_bands=nb, rows_per_band=r, num_perm=nb * r, buffer_size=buffer_size,
                          storage=s2, packed_ingest=False)
    with pytest.raises(ValueError, match="Cannot index zero vector"):
        idx2.index(ids, bad)
    O.index_literal(w2, ids[:100], bad[:100], P, dim, buffer_size)
    flushed = [op for batch in s2.batches for op in batch] + list(idx2._buffer)
    assert flushed == [op for batch in w2.batches for op in batch]
    assert [len(b) for b in s2.batches] == [len(b) for b in w2.batches][:len(s2.batches)]


def test_in_memory_store_takes_any_key_and_id_types():
    """`batch_add` keeps buckets under (band, key bytes): what LSHRS sends goes in as it is; other byte-likes / NumPy integers
    are normalised - same buckets, same key texts."""
    a, b = InMemoryStorage(), InMemoryStorage(record_batches=False)
    ops = [(0, b"\x01\x02", 5), (1, b"\x00\x00", 6), (0, b"\x01\x02", 7)]
    a.batch_add(ops)
    b.batch_add([(np.int64(0), bytearray(b"\x01\x02"), np.int64(5)), (1, memoryview(b"\x00\x00"), 6), (0, b"\x01\x02", np.int32(7))])
    assert a.bucket_contents() == b.bucket_contents() == {"lsh:0:bucket:0102": {5, 7}, "lsh:1:bucket:0000": {6}}
    assert a.batches == [ops] and b.batches == [] and b.get_bucket(0, b"\x01\x02") == {5, 7}
    assert all(type(i) is int for i in b.get_bucket(0, bytearray(b"\x01\x02")))
