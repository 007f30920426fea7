"""Packed-key -> storage-op path: grouping and writers (CPU), hex kernel (GPU)."""

from __future__ import annotations

import contextlib

import numpy as np
import pytest

from lshrs_amd import InMemoryStorage, LSHRS, RedisPackedWriter, group_by_bucket
from tests._doubles import OracleBackedHasher, make_cpu_lshrs


def ops_from_keys(ids, keys):
    return [(b, keys[j, b].tobytes(), int(ids[j])) for j in range(len(ids)) for b in range(keys.shape[1])]


def test_groups_cover_exactly_the_reference_operations():
    rng = np.random.default_rng(0)
    keys = rng.integers(0, 4, size=(500, 6, 2), dtype=np.uint8)     # few distinct values: crowded buckets
    ids = rng.permutation(10_000)[:500]
    got = set()
    for band, key, members in group_by_bucket(ids, keys):
        assert len(set(members.tolist())) == len(members)
        got.update((band, key, int(m)) for m in members)
    assert got == set(ops_from_keys(ids, keys))
    with pytest.raises(ValueError):
        list(group_by_bucket(ids[:10], keys))


def test_packed_ingest_builds_the_same_buckets(monkeypatch):
    rng = np.random.default_rng(1)
    data = rng.standard_normal((700, 32)).astype(np.float32)
    data[300:340] = data[:40] + 1e-3 * rng.standard_normal((40, 32)).astype(np.float32)   # shared buckets
    plain, packed = InMemoryStorage(), InMemoryStorage()
    make_cpu_lshrs(monkeypatch, dim=32, num_bands=4, rows_per_band=4, num_perm=16, storage=plain).index(list(range(700)), data)
    idx = make_cpu_lshrs(monkeypatch, dim=32, num_bands=4, rows_per_band=4, num_perm=16, storage=packed,
                         packed_ingest=True, buffer_size=400)
    idx.ingest(9999, data[0])                       # something buffered: must be flushed before the packed batch
    idx.index(list(range(700)), data)
    assert packed.batches and packed.batches[0][0][2] == 9999
    assert [n for n, _ in packed.packed_batches] == [700]               # the whole batch as one bucket CSR
    expect = plain.bucket_contents()
    for b, kb, i in packed.batches[0]:
        expect.setdefault(packed.bucket_key(b, kb), set()).add(i)
    assert packed.bucket_contents() == expect
    assert idx.get_top_k(data[5], topk=1) == [5] or 9999 in idx.get_top_k(data[0], topk=2)
    # error timing: rows before the bad one are stored, then the same error
    bad = data[:10].copy()
    bad[6] = 0
    store = InMemoryStorage()
    with pytest.raises(ValueError, match="zero vector"):
        make_cpu_lshrs(monkeypatch, dim=32, num_bands=4, rows_per_band=4, num_perm=16, storage=store,
                       packed_ingest=True).index(list(range(10)), bad)
    assert {i for s in store.bucket_contents().values() for i in s} == set(range(6))


class FakePipeline:
    def __init__(self, log):
        self.log = log

    def sadd(self, name, *members):
        self.log.append((name, members))


class FakeRedisStorage:
    """The two members of the reference's RedisStorage the writer uses (redis.py:187, :508)."""
    prefix = "lsh"

    def __init__(self):
        self.commands = []

    def bucket_key(self, band_id, hash_val):
        return f"{self.prefix}:{band_id}:bucket:{hash_val.hex()}"

    @contextlib.contextmanager
    def pipeline(self):
        yield FakePipeline(self.commands)

    def get_bucket(self, band_id, hash_val):
        return {1, 2}


def test_redis_writer_sends_the_same_members_in_fewer_commands():
    rng = np.random.default_rng(2)
    keys = rng.integers(0, 3, size=(400, 4, 1), dtype=np.uint8)
    ids = np.arange(400)
    fake = FakeRedisStorage()
    writer = RedisPackedWriter(fake, max_members_per_command=50)
    n = writer.batch_add_packed(ids, keys)
    assert n == len(fake.commands) < 400 * 4
    assert all(len(m) <= 50 for _, m in fake.commands)
    sent = {(name, m) for name, members in fake.commands for m in members}
    want = {(fake.bucket_key(b, k), i) for b, k, i in ops_from_keys(ids, keys)}   # what one SADD per op would add
    assert sent == want
    assert writer.get_bucket(0, b"\x00") == {1, 2}          # everything else is delegated


def test_default_ingest_mode_is_auto(monkeypatch):
    """INTEGRATION option B unchanged (no `packed_ingest` argument) takes the array path where that is
    safe - the storage can take a bucket CSR, or is the reference's RedisStorage (pipeline() + bucket_key(): wrapped in
    RedisPackedWriter) - and the reference's operation tuples otherwise; small batches keep the tuples.  Same buckets."""
    rng = np.random.default_rng(4)
    data = rng.standard_normal((3000, 32)).astype(np.float32)
    kw = dict(dim=32, num_bands=4, rows_per_band=4, num_perm=16)
    tuples, auto = InMemoryStorage(), InMemoryStorage()
    make_cpu_lshrs(monkeypatch, storage=tuples, packed_ingest=False, **kw).index(list(range(3000)), data)
    idx = LSHRS(storage=auto, hasher=OracleBackedHasher(4, 4, 32, 42), **kw)        # no packed_ingest argument
    assert idx._packed_ingest == "auto"
    from lshrs_amd import core, packed_ops
    monkeypatch.setattr(core, "_bucket_csr", lambda ids, keys: packed_ops._csr_host(
        np.ascontiguousarray(np.asarray(ids, dtype=np.int64)), np.ascontiguousarray(keys, dtype=np.uint8)))
    idx.index(list(range(3000)), data)                                              # 12 000 operations: one bucket CSR
    assert [n for n, _ in auto.packed_batches] == [3000] and not auto.batches
    assert auto.bucket_contents() == tuples.bucket_contents()
    idx.index([5000, 5001], data[:2])                                               # 8 operations: the reference's tuples
    assert [len(b) for b in auto.batches] == [8]
    # the reference's RedisStorage interface: one SADD per bucket through its own pipeline
    fake = FakeRedisStorage()
    fake.batch_add = lambda ops: fake.commands.extend((fake.bucket_key(b, k), (i,)) for b, k, i in ops)
    r = LSHRS(storage=fake, hasher=OracleBackedHasher(4, 4, 32, 42), **kw)
    r.index(list(range(3000)), data)
    sent = {(name, int(m)) for name, members in fake.commands for m in members}
    assert sent == {(name, m) for name, ms in tuples.bucket_contents().items() for m in ms}
    assert len(fake.commands) < 3000 * 4 // 2                                       # (buckets, not operations)
    # a storage with neither interface (the reference's MockStorage): tuples, whatever the size
    class Plain:
        def __init__(self):
            self.ops = []
        def batch_add(self, ops):
            self.ops.extend(ops)
    plain = Plain()
    LSHRS(storage=plain, hasher=OracleBackedHasher(4, 4, 32, 42), **kw).index(list(range(3000)), data)
    assert len(plain.ops) == 12_000
    with pytest.raises(ValueError):
        LSHRS(storage=plain, packed_ingest="yes", **kw)


def test_disjoint_segments_need_no_dedupe_and_overlapping_ones_get_it():
    """sequential ingest leaves segments with disjoint id ranges - a lookup then concatenates their members
    without sorting every (query, band, member) pair; ids indexed twice (overlapping ranges) still count once per band."""
    from lshrs_amd.packed_ops import _csr_host

    rng = np.random.default_rng(6)
    keys = rng.integers(0, 3, size=(600, 4, 1), dtype=np.uint8)
    store = InMemoryStorage()
    store.batch_add_csr(_csr_host(np.arange(0, 300, dtype=np.int64), keys[:300]))
    store.batch_add_csr(_csr_host(np.arange(300, 600, dtype=np.int64), keys[300:]))
    q, m = store.get_buckets_many(keys[:5])
    want = sorted((qi, b, i) for qi in range(5) for b in range(4) for i in range(600) if keys[i, b, 0] == keys[qi, b, 0])
    assert sorted(zip(q.tolist(), m.tolist())) == sorted((qi, i) for qi, _, i in want)
    store.batch_add_csr(_csr_host(np.arange(100, 200, dtype=np.int64), keys[100:200]))      # the same ids again
    q2, m2 = store.get_buckets_many(keys[:5])
    assert sorted(zip(q2.tolist(), m2.tolist())) == sorted(zip(q.tolist(), m.tolist()))


@pytest.mark.gpu
def test_hex_kernel_equals_bytes_hex():
    import torch

    from lshrs_amd import hex_keys
    from lshrs_amd.packed_ops import hex_keys_device

    rng = np.random.default_rng(3)
    for shape in ((1000, 16, 2), (77, 3, 1), (5, 16, 4), (1, 1, 3), (0, 4, 2)):
        keys = rng.integers(0, 256, size=shape, dtype=np.uint8)
        hx = hex_keys(keys)
        assert hx.shape == shape[:2] and hx.dtype == np.dtype(f"S{2 * shape[2]}")
        for i in range(0, shape[0], max(1, shape[0] // 7)):
            for b in range(shape[1]):
                assert hx[i, b] == keys[i, b].tobytes().hex().encode()
    every = np.arange(256, dtype=np.uint8).reshape(16, 16, 1)
    assert [h.decode() for h in hex_keys(every).reshape(-1)] == [f"{v:02x}" for v in range(256)]
    dev = hex_keys_device(torch.from_numpy(every).cuda())
    assert dev.shape == (16, 16, 2) and dev.is_cuda


@pytest.mark.gpu
def test_device_bucket_csr_equals_host_grouping():
    """lshrs_bucket_histogram_u8 / lshrs_bucket_scatter_u8 (counting sort per band) against the NumPy grouping: same
    buckets, same member sets, offsets consistent; keys of 1 and 2 bytes on the device, wider ones on the host."""
    import torch

    from lshrs_amd.packed_ops import _csr_host, bucket_csr

    rng = np.random.default_rng(17)
    for (n, nb, bb, hi) in ((200_000, 16, 2, 256), (50_000, 16, 1, 16), (1, 3, 2, 256), (4097, 5, 2, 4), (3000, 16, 4, 256),
                            (120_000, 16, 4, 3), (777, 2, 6, 2), (500, 3, 7, 2)):
        keys = rng.integers(0, hi, size=(n, nb, bb), dtype=np.uint8)
        ids = rng.permutation(10 * n + 5)[:n].astype(np.int64)
        want = _csr_host(ids, keys)
        for source in (keys, torch.from_numpy(keys).cuda()):
            got = bucket_csr(ids, source)
            assert len(got) == len(want) and got.vectors == n and got.band_bytes == bb
            assert np.array_equal(got.bands, want.bands) and np.array_equal(got.key_bytes, want.key_bytes)
            assert np.array_equal(got.offsets, want.offsets) and got.members.shape == (n * nb,)
            for g in rng.choice(len(want), size=min(len(want), 300), replace=False):
                lo, hi_ = want.offsets[g], want.offsets[g + 1]
                assert sorted(got.members[lo:hi_].tolist()) == sorted(want.members[lo:hi_].tolist())
            assert np.array_equal(np.sort(got.members), np.sort(want.members))
    assert len(bucket_csr(np.empty(0, np.int64), np.empty((0, 16, 2), np.uint8))) == 0


def test_redis_writer_takes_a_bucket_csr():
    """``RedisPackedWriter.batch_add_csr``: one ``SADD`` per bucket (split at max_members), the reference's key text
    (lshrs/storage/redis.py:187-225), exactly the members one SADD per operation would have added."""
    from lshrs_amd.packed_ops import _csr_host

    rng = np.random.default_rng(9)
    keys = rng.integers(0, 3, size=(400, 4, 2), dtype=np.uint8)
    ids = rng.permutation(5000)[:400].astype(np.int64)
    fake = FakeRedisStorage()
    writer = RedisPackedWriter(fake, max_members_per_command=50)
    n = writer.batch_add_csr(_csr_host(ids, keys))
    assert n == len(fake.commands) and all(len(m) <= 50 for _, m in fake.commands)
    sent = {(name, m) for name, members in fake.commands for m in members}
    assert sent == {(fake.bucket_key(b, k), i) for b, k, i in ops_from_keys(ids, keys)}
    assert all(isinstance(m, int) for _, members in fake.commands for m in members)


def test_array_segments_are_compacted_and_stay_equal_to_the_tuple_store():
    """Many small packed batches: past `compact_above` segments a lookup folds them into one (merge_csr); contents,
    single lookups, batched lookups and deletions stay equal to a store fed the same operations as tuples."""
    from lshrs_amd import InMemoryStorage
    from lshrs_amd.packed_ops import _csr_host, merge_csr

    rng = np.random.default_rng(11)
    a, b = InMemoryStorage(), InMemoryStorage()
    a.compact_above = 6
    next_id = 0
    for batch in range(25):
        n = int(rng.integers(1, 400))
        keys = rng.integers(0, 6, size=(n, 5, 2), dtype=np.uint8)          # few distinct keys: crowded, shared buckets
        ids = np.arange(next_id, next_id + n, dtype=np.int64) * 7 + 3
        next_id += n
        a.batch_add_csr(_csr_host(ids, keys))
        b.batch_add([(band, keys[i, band].tobytes(), int(ids[i])) for i in range(n) for band in range(5)])
        if batch % 6 == 5:
            gone = rng.choice(next_id, size=20, replace=False) * 7 + 3
            a.remove_indices(gone.tolist()); b.remove_indices(gone.tolist())
        q = rng.integers(0, 6, size=(9, 5, 2), dtype=np.uint8)
        for qi in range(9):
            for band in range(5):
                assert a.get_bucket(band, q[qi, band].tobytes()) == b.get_bucket(band, q[qi, band].tobytes())
        qa, ma = a.get_buckets_many(q)
        qb, mb = b.get_buckets_many(q)
        assert sorted(zip(qa.tolist(), ma.tolist())) == sorted(zip(qb.tolist(), mb.tolist()))
        assert len(a._segments) <= a.compact_above + 1
    assert a.bucket_contents() == b.bucket_contents()
    # the merge on its own: union of buckets, members concatenated
    segs = [_csr_host(np.arange(50, dtype=np.int64) + 100 * s, rng.integers(0, 3, size=(50, 2, 1), dtype=np.uint8)) for s in range(4)]
    merged = merge_csr(segs)
    want = {}
    for sgm in segs:
        for g in range(len(sgm)):
            want.setdefault(int(sgm.codes[g]), []).extend(sgm.members[sgm.offsets[g]:sgm.offsets[g + 1]].tolist())
    got = {int(merged.codes[g]): merged.members[merged.offsets[g]:merged.offsets[g + 1]].tolist() for g in range(len(merged))}
    assert got == want and merged.vectors == 200 and np.all(np.diff(merged.codes) > 0)


def test_array_buckets_are_sets_like_the_references_redis_sets():
    """an id indexed twice - in one batch, in two batches, or once as an op tuple and once in an array
    segment - is in its bucket ONCE (SADD, lshrs/storage/redis.py:408-416), so a query counts it once per band
    (lshrs/core/main.py:1101-1109).  The array-fed store against the tuple-fed one."""
    from lshrs_amd.packed_ops import _csr_host, dedupe_csr, merge_csr
    from lshrs_amd.storage import InMemoryStorage

    rng = np.random.default_rng(11)
    nb, bb = 4, 1
    keys = rng.integers(0, 3, (12, nb, bb)).astype(np.uint8)
    ids = np.array([5, 5, 7, 9, 9, 9, 1, 2, 3, 4, 6, 8], dtype=np.int64)
    keys[1], keys[4], keys[5] = keys[0], keys[3], keys[3]                # a re-indexed id carries the same vector
    tuple_store, array_store, mixed = InMemoryStorage(), InMemoryStorage(), InMemoryStorage()

    def feed_tuples(store, id_arr, key_arr):
        store.batch_add([(b, key_arr[i, b].tobytes(), int(id_arr[i])) for i in range(len(id_arr)) for b in range(nb)])

    for rep in range(3):                                                 # ... and the whole batch three times over
        feed_tuples(tuple_store, ids, keys)
        array_store.batch_add_csr(_csr_host(ids, keys))                  # (the host builder: no GPU in this test)
        (feed_tuples if rep == 1 else lambda s, i, k: s.batch_add_csr(_csr_host(i, k)))(mixed, ids, keys)
    assert array_store.bucket_contents() == tuple_store.bucket_contents() == mixed.bucket_contents()
    for store in (array_store, mixed):
        for seg in store._segments:
            assert seg.distinct
            for g in range(len(seg)):
                mem = seg.members[seg.offsets[g]:seg.offsets[g + 1]]
                assert len(set(mem.tolist())) == len(mem)
    q = keys[[0, 3, 6, 7]]
    want_q, want_m = tuple_store.get_buckets_many(q)
    want = sorted(zip(want_q.tolist(), want_m.tolist()))
    for store in (array_store, mixed):
        gq, gm = store.get_buckets_many(q)
        assert sorted(zip(gq.tolist(), gm.tolist())) == want
        # every (query, member) pair at most once per band: never more than nb times in all
        pairs, counts = np.unique(np.stack([gq, gm], axis=1), axis=0, return_counts=True)
        assert counts.max() <= nb
    # folding the segments keeps the sets
    merged = merge_csr(list(array_store._segments))
    assert merged.distinct and int(np.diff(merged.offsets).sum()) == sum(len(v) for v in tuple_store.bucket_contents().values())
    assert dedupe_csr(merged) is merged


def test_reindexed_ids_rank_like_the_tuple_fed_store():
    """The ordered candidates of a query (collision count, then id: lshrs/core/main.py:614) with ids indexed several
    times: array-fed == tuple-fed, and no candidate is lost (the packed sort key of `_ordered_candidates_arrays` went
    negative when a count exceeded num_bands)."""
    from lshrs_amd.core import LSHRS
    from lshrs_amd.packed_ops import _csr_host
    from lshrs_amd.storage import InMemoryStorage

    rng = np.random.default_rng(12)
    nb, bb = 4, 1
    keys = rng.integers(0, 2, (3, nb, bb)).astype(np.uint8)
    ids = np.array([10, 11, 12], dtype=np.int64)
    a, t = InMemoryStorage(), InMemoryStorage()
    for _ in range(3):
        a.batch_add_csr(_csr_host(ids, keys))
        t.batch_add([(b, keys[i, b].tobytes(), int(ids[i])) for i in range(3) for b in range(nb)])
    one = _csr_host(np.array([5, 5, 7], dtype=np.int64), keys)           # an id twice inside one batch
    from lshrs_amd.packed_ops import dedupe_csr
    a.batch_add_csr(one)
    t.batch_add([(b, keys[i, b].tobytes(), int(v)) for i, v in enumerate((5, 5, 7)) for b in range(nb)])

    class _Shell(LSHRS):                                                  # only the candidate ordering is under test
        def __init__(self, storage):
            self._storage = storage

    got = _Shell(a)._ordered_candidates_many(keys[:2])
    want = _Shell(t)._ordered_candidates_many(keys[:2])
    assert got == want and all(len(g) > 0 for g in got)
    assert {10, 5} <= set(got[0])


def test_redis_writer_opens_a_new_pipeline_every_buffer_size_members():
    """the default `packed_ingest="auto"` wraps the reference's RedisStorage in this writer - a 1 M x 16 band
    batch must not become ONE pipeline of 16 M members.  A pipeline is executed every `flush_members` (= `LSHRS.buffer_size`)
    members, as the reference's index() flushes its buffer every `buffer_size` operations (lshrs/core/main.py:1131-1143)."""
    from lshrs_amd.packed_ops import _csr_host

    class CountingStorage(FakeRedisStorage):
        def __init__(self):
            super().__init__()
            self.per_pipeline = []

        @contextlib.contextmanager
        def pipeline(self):
            log = []
            yield FakePipeline(log)
            self.per_pipeline.append(sum(len(m) for _, m in log))      # "executed" on exit
            self.commands.extend(log)

    rng = np.random.default_rng(12)
    keys = rng.integers(0, 4, size=(3000, 8, 1), dtype=np.uint8)         # 24 000 operations in 32 crowded buckets
    ids = np.arange(3000, dtype=np.int64)
    want = {(FakeRedisStorage().bucket_key(b, k), i) for b, k, i in ops_from_keys(ids, keys)}
    for send in ("csr", "packed"):
        fake = CountingStorage()
        writer = RedisPackedWriter(fake, flush_members=1000)
        n = writer.batch_add_csr(_csr_host(ids, keys)) if send == "csr" else writer.batch_add_packed(ids, keys)
        assert n == len(fake.commands)
        assert writer.pipelines == len(fake.per_pipeline) == 24 and max(fake.per_pipeline) <= 1000 and sum(fake.per_pipeline) == 24_000
        assert {(name, m) for name, members in fake.commands for m in members} == want
    # LSHRS hands its buffer_size to the writer it builds around such a storage
    idx = LSHRS(dim=8, num_perm=8, num_bands=4, rows_per_band=2, storage=CountingStorage(), buffer_size=777, packed_ingest=True)
    sink = idx._packed_sink(10 ** 6)
    assert isinstance(sink, RedisPackedWriter) and sink.flush_members == 777
