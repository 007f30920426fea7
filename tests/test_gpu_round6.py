"""Round 6: the advisor's findings on the pipelined ingest (the sink outside the hasher's lock, a stream that ends when the
unit does, chained exceptions, the device -> pinned-host copy at any address) - on the GPU, through the public entry points."""

from __future__ import annotations

import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_copy_to_host_at_any_address_and_length():
    """`lshrs_copy_to_host_u8` with source / destination at odd offsets and odd lengths - 16 bytes per lane where the
    two sit alike modulo 16 (head and tail byte by byte), bytes over the whole grid where they do not."""
    import torch

    from lshrs_amd import _native

    lib = _native.load()
    dev = torch.device("cuda", 0)
    src = torch.arange(0, 3_000_000, device=dev, dtype=torch.int32).view(torch.uint8)          # 12 MB of distinct bytes
    host = torch.zeros(src.numel() + 64, dtype=torch.uint8).pin_memory()
    stream = torch.cuda.current_stream(dev)
    for s_off, d_off, n in [(0, 0, 1), (0, 0, 15), (0, 0, 17), (0, 0, 257), (0, 0, 4099), (1, 1, 4099), (3, 3, 1_000_003),
                            (5, 21, 100_001), (1, 0, 4099), (7, 2, 1_000_000), (0, 9, 33), (13, 13, 5), (0, 0, 12_000_000),
                            (15, 15, 2), (8, 8, 7), (4, 12, 70_001)]:
        host.zero_()
        _native.check(lib.lshrs_copy_to_host_u8(src.data_ptr() + s_off, host.data_ptr() + d_off, n, stream.cuda_stream),
                      "lshrs_copy_to_host_u8")
        stream.synchronize()
        got = host.numpy()
        want = src[s_off:s_off + n].cpu().numpy()
        assert np.array_equal(got[d_off:d_off + n], want), (s_off, d_off, n)
        assert not got[:d_off].any() and not got[d_off + n:d_off + n + 40].any(), (s_off, d_off, n)      # nothing outside the range


class _SlowStore:
    """An array-taking store whose `batch_add_csr` blocks until the test lets it go."""

    def __init__(self):
        from lshrs_amd import InMemoryStorage

        self.inner = InMemoryStorage()
        self.entered = threading.Event()
        self.release = threading.Event()
        self.calls = 0

    def batch_add_csr(self, csr):
        self.calls += 1
        self.entered.set()
        assert self.release.wait(60)
        self.inner.batch_add_csr(csr)

    def __getattr__(self, item):
        return getattr(self.inner, item)


def test_queries_and_ingests_go_on_while_a_unit_is_being_stored():
    """the device sink - finish the chunk's grouping, wait for the previous unit, write to the store - runs
    WITHOUT the hasher's lock: while `index()` sits in a slow store, `get_top_k`, `ingest` and a `hash_device` on the same
    hasher return."""
    import torch

    from lshrs_amd import LSHRS

    rng = np.random.default_rng(3)
    n, dim = 140_000, 128
    data = rng.standard_normal((n, dim)).astype(np.float32)
    store = _SlowStore()
    idx = LSHRS(dim=dim, num_perm=64, storage=store, packed_ingest=True)
    store.release.set()
    idx.index(np.arange(1000), data[:1000])             # (warm: workspace, windows; the store open)
    store.release.clear()
    store.entered.clear()
    done = threading.Event()
    err = []

    def run():
        try:
            idx.index(np.arange(1000, n), data[1000:])
        except BaseException as exc:  # noqa: BLE001
            err.append(exc)
        done.set()

    t = threading.Thread(target=run)
    t.start()
    assert store.entered.wait(60)                       # the first chunk is in the store's hands - and stays there
    t0 = time.perf_counter()
    got = idx.get_top_k(data[5], topk=3)                # hash_one -> the hasher's lock: free
    keys = idx._hasher.hash_device(torch.from_numpy(data[:512]).cuda())
    idx.ingest(10**9, data[7])
    waited = time.perf_counter() - t0
    assert 5 in got and keys.shape == (512, 16, 1) and not done.is_set() and waited < 20
    store.release.set()
    assert done.wait(120) and not err, err
    t.join()
    assert idx.get_top_k(data[70_000], topk=1) == [70_000]


def test_a_unit_that_ends_early_ends_the_stream():
    """a zero vector in the first chunk of a long unit - the rows in front of it are stored, the reference's
    error is raised, and the chunks behind it are neither copied nor hashed (the sink returns False)."""
    from lshrs_amd import LSHRS, InMemoryStorage

    rng = np.random.default_rng(4)
    n, dim = 700_000, 64
    data = rng.standard_normal((n, dim)).astype(np.float32)
    data[1234] = 0
    store = InMemoryStorage()
    idx = LSHRS(dim=dim, num_perm=64, storage=store, packed_ingest=True)
    launches = []
    launch = idx._hasher._hash_device_async_locked
    idx._hasher._hash_device_async_locked = lambda x, out, fl: (launches.append(int(x.shape[0])), launch(x, out, fl))[1]
    with pytest.raises(ValueError, match="Cannot index zero vector"):
        idx.index(np.arange(n), data)
    assert 1 <= len(launches) <= 3 < -(-n // 131_072), launches          # (six chunks in the unit: the stream stopped behind the second)
    stored = sum(v for v, _ in store.packed_batches)
    assert stored == 1234
    assert idx.get_top_k(data[1000], topk=1) == [1000] and idx.get_top_k(data[5000], topk=1) != [5000]


def test_loader_errors_keep_their_cause_and_bad_rows_come_first():
    """`create_signatures` - the loader raises after handing over a unit with a bad row: the bad row's error
    (first in row order) is raised, the loader's exception chained behind it; a loader error alone is raised as it is, after the
    units handed over were stored."""
    from lshrs_amd import LSHRS, InMemoryStorage

    rng = np.random.default_rng(5)
    dim, per = 64, 40_000
    blocks = [rng.standard_normal((per, dim)).astype(np.float32) for _ in range(3)]

    class Boom(RuntimeError):
        pass

    def loader(bad_row):
        for i, b in enumerate(blocks):
            if i == 2:
                raise Boom("the source went away")
            if bad_row and i == 0:
                b = b.copy()
                b[77] = 0
            yield np.arange(i * per, (i + 1) * per), b

    store = InMemoryStorage()
    idx = LSHRS(dim=dim, num_perm=64, storage=store, packed_ingest=True)
    with pytest.raises(Boom):
        idx.create_signatures(format="batches", batches=loader(False))
    assert sum(v for v, _ in store.packed_batches) == 2 * per            # both units handed over in front of the failure are stored
    store2 = InMemoryStorage()
    idx2 = LSHRS(dim=dim, num_perm=64, storage=store2, packed_ingest=True)
    with pytest.raises(ValueError, match="Cannot index zero vector") as info:
        idx2.create_signatures(format="batches", batches=loader(True))
    assert sum(v for v, _ in store2.packed_batches) == 77
    cause = info.value.__cause__ or info.value.__context__
    assert cause is None or isinstance(cause, Boom)       # (the bad row may surface through `ingest.failed` before the loader is asked again)


def test_one_query_per_call_follows_the_store_through_its_changes():
    """Round 6: `get_top_k` keeps the device descriptor of the store's segments beside the store's change token - every way the
    store can change (another `index`, `delete`, `clear`, op-tuple buckets appearing, another store) is seen by the next call."""
    from lshrs_amd import LSHRS, InMemoryStorage

    rng = np.random.default_rng(12)
    dim = 64
    data = rng.standard_normal((6000, dim)).astype(np.float32)
    store = InMemoryStorage()
    idx = LSHRS(dim=dim, num_perm=64, storage=store, packed_ingest=True)
    idx.index(np.arange(2000), data[:2000])
    assert idx.get_top_k(data[10], topk=1) == [10] and idx.get_top_k(data[10], topk=1) == [10]
    assert idx._one_table is not None
    kept = idx._one_table[0]
    assert idx.get_top_k(data[3000], topk=1) != [3000]
    assert idx._one_table[0] == kept                                 # (the same entry served the calls in between)
    idx.index(np.arange(2000, 4000), data[2000:4000])                # another segment
    assert idx.get_top_k(data[3000], topk=1) == [3000] and idx._one_table[0] != kept
    idx.delete(3000)
    assert 3000 not in idx.get_top_k(data[3000], topk=5)
    idx.ingest(5000, data[5000])                                     # an op-tuple bucket: the store answers through get_bucket again
    idx.flush()
    assert idx.get_top_k(data[5000], topk=1) == [5000]
    assert idx.get_top_k(data[10], topk=1) == [10]
    store.clear()
    assert idx.get_top_k(data[10], topk=1) == []
    idx.index(np.arange(100), data[:100])
    assert idx.get_top_k(data[10], topk=1) == [10]
    other = InMemoryStorage()
    idx2 = LSHRS(dim=dim, num_perm=64, storage=other, packed_ingest=True)
    idx2._hasher.projections = idx._hasher.projections
    idx2.index(np.arange(4000, 4100), data[4000:4100])
    assert idx2.get_top_k(data[4050], topk=1) == [4050] and idx.get_top_k(data[4050], topk=1) != [4050]


def test_vectors_that_live_on_the_gpu_are_indexed_where_they_are():
    """Round 6: `index(ids, x)` with a torch tensor on the GPU - the same buckets as the same rows from host memory (whole and
    ragged chunks, a strided view, float64), the reference's errors at the reference's rows (a zero vector: the rows in front of
    it stored; a negative id; wrong shapes), and a store without arrays gets the host form."""
    import torch

    from lshrs_amd import LSHRS, InMemoryStorage

    rng = np.random.default_rng(31)
    n, dim = 300_001, 64
    data = rng.standard_normal((n, dim)).astype(np.float32)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(data).to(dev)
    ids = rng.permutation(10 * n)[:n].astype(np.int64)

    def built(vectors, idv=ids, **kw):
        st = InMemoryStorage()
        idx = LSHRS(dim=dim, num_perm=64, storage=st, packed_ingest=True, **kw)
        idx.index(idv, vectors)
        return idx, st

    _, want = built(data)
    idx, got = built(x)
    assert got.bucket_contents() == want.bucket_contents()
    assert sum(v for v, _ in got.packed_batches) == n
    assert idx.get_top_k(data[777], topk=1) == [int(ids[777])]
    assert idx.get_top_k(x[777], topk=1) == [int(ids[777])]                         # one vector on the GPU: the single-vector calls too
    idx.ingest(10 ** 12, x[5])
    wide = torch.zeros(n, dim + 8, device=dev)
    wide[:, :dim] = x
    assert built(wide[:, :dim])[1].bucket_contents() == want.bucket_contents()          # rows of a wider matrix
    assert built(x[:5000].double(), ids[:5000])[1].bucket_contents() == built(data[:5000], ids[:5000])[1].bucket_contents()
    # the reference's errors, at its rows
    bad = x.clone()
    bad[200_123] = 0
    st = InMemoryStorage()
    idx = LSHRS(dim=dim, num_perm=64, storage=st, packed_ingest=True)
    with pytest.raises(ValueError, match="Cannot index zero vector"):
        idx.index(ids, bad)
    assert sum(v for v, _ in st.packed_batches) == 200_123
    neg = ids.copy()
    neg[150_000] = -5
    st2 = InMemoryStorage()
    with pytest.raises(ValueError, match="index must be non-negative"):
        LSHRS(dim=dim, num_perm=64, storage=st2, packed_ingest=True).index(neg, x)
    assert sum(v for v, _ in st2.packed_batches) == 150_000
    with pytest.raises(ValueError, match="Vectors must have shape"):
        idx.index(ids, x[:, :10])
    with pytest.raises(ValueError, match="Number of vectors does not match"):
        idx.index(ids[:10], x)
    # a loader that yields device tensors (`create_signatures(format="batches")`): the pipelined ingest takes them too
    st3 = InMemoryStorage()
    idx3 = LSHRS(dim=dim, num_perm=64, storage=st3, packed_ingest=True)
    idx3.create_signatures(format="batches", batches=((ids[lo:lo + 100_000], x[lo:lo + 100_000]) for lo in range(0, n, 100_000)))
    assert st3.bucket_contents() == want.bucket_contents()
    # a store that only takes operation tuples: the same rows, from the host
    class Plain:
        def __init__(self):
            self.ops = []

        def batch_add(self, ops):
            self.ops.extend(ops)

        def get_bucket(self, b, k):
            return set()

    p1, p2 = Plain(), Plain()
    LSHRS(dim=dim, num_perm=64, storage=p1).index(ids[:3000], x[:3000])
    LSHRS(dim=dim, num_perm=64, storage=p2).index(ids[:3000], data[:3000])
    assert p1.ops == p2.ops and len(p1.ops) == 3000 * 16
