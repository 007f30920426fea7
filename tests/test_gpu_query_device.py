"""The candidate path of ``query_many`` on the device (round 6): bucket lookup, collision
counting, candidate order, rerank, top-p / top-k cut - kernels of csrc/query.hip through the C ABI - against the
reference's flow restated literally (oracle.query_literal: lshrs/core/main.py:524-658, :1088-1111) and against plain
Python dict counting."""

from __future__ import annotations

import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _literal_lists(segments, keys, nb, bb):
    """Per query: [(id, collisions)] ordered by (-collisions, id) - dict counting over the buckets the keys select, an id
    once per band however many segments list it there."""
    out = []
    for qi in range(keys.shape[0]):
        counts = {}
        for b in range(nb):
            code = (b << (8 * bb)) | int.from_bytes(keys[qi, b].tobytes(), "little")
            members = set()
            for seg in segments:
                g = int(np.searchsorted(seg.codes, code))
                if g < len(seg) and int(seg.codes[g]) == code:
                    members.update(seg.members[seg.offsets[g]:seg.offsets[g + 1]].tolist())
            for m in members:
                counts[m] = counts.get(m, 0) + 1
        out.append(sorted(counts.items(), key=lambda kv: (-kv[1], kv[0])))
    return out


def _random_segments(rng, nb, bb, nseg, n_rows, key_space, id_hi):
    """`nseg` BucketCSR segments over a small key space (real collisions), ids overlapping between segments."""
    from lshrs_amd.packed_ops import _csr_host

    segs = []
    for _ in range(nseg):
        ids = rng.choice(id_hi, size=n_rows, replace=False).astype(np.int64)
        keys = np.zeros((n_rows, nb, bb), dtype=np.uint8)
        vals = rng.integers(0, key_space, size=(n_rows, nb))
        for j in range(bb):
            keys[:, :, j] = (vals >> (8 * j)) & 0xFF
        csr = _csr_host(ids, keys)
        csr.distinct = True
        segs.append(csr)
    return segs


@pytest.mark.parametrize("nb,bb,nseg,key_space", [(16, 2, 3, 40), (1, 1, 1, 5), (3, 1, 2, 7), (20, 2, 4, 300), (16, 4, 2, 25),
                                                   (33, 1, 2, 4), (128, 1, 5, 64), (8, 6, 2, 12)])
def test_lookup_and_collide_equal_dict_counting(nb, bb, nseg, key_space):
    import torch

    from lshrs_amd import _query_device as qd

    rng = np.random.default_rng(nb * 131 + bb)
    segs = _random_segments(rng, nb, bb, nseg, 400, key_space, 600)        # ids overlap between segments: the same id twice in a bucket
    nq = 257
    vals = rng.integers(0, key_space + 3, size=(nq, nb))                    # (+3: some keys no bucket has)
    keys = np.zeros((nq, nb, bb), dtype=np.uint8)
    for j in range(bb):
        keys[:, :, j] = (vals >> (8 * j)) & 0xFF
    dev = torch.device("cuda", 0)
    mirror = qd.DeviceBuckets()
    desc, n, max_id = mirror.table(segs, dev)
    assert n == nseg and max_id < 600
    lists = qd.candidates_from_index(torch.from_numpy(keys).to(dev), desc, n, max_id, want_hits=True)
    off, cnt = lists.pair_off.cpu().numpy(), lists.ucount.cpu().numpy()
    ids, hits = lists.cand_ids.cpu().numpy(), lists.hits.cpu().numpy()
    want = _literal_lists(segs, keys, nb, bb)
    assert (vals >= key_space).any()                                         # keys that select no bucket
    assert nb == 1 or any(h > 1 for w in want for _, h in w)                 # an id in several bands
    for qi in range(nq):
        got = list(zip(ids[off[qi]:off[qi] + cnt[qi]].tolist(), hits[off[qi]:off[qi] + cnt[qi]].tolist()))
        assert got == want[qi], qi
    # the same lists from pairs handed over flat (a store with get_bucket only); every (member, band) pair twice: sets
    ms, bs, po = [], [], [0]
    for qi in range(nq):
        n_pairs = 0
        for b in range(nb):
            code = (b << (8 * bb)) | int.from_bytes(keys[qi, b].tobytes(), "little")
            for seg in segs:
                g = int(np.searchsorted(seg.codes, code))
                if g < len(seg) and int(seg.codes[g]) == code:
                    mem = seg.members[seg.offsets[g]:seg.offsets[g + 1]]
                    ms.append(mem)
                    bs.append(np.full(len(mem), b, np.int32))
                    n_pairs += len(mem)
        po.append(po[-1] + n_pairs)
    flat = qd.candidates_from_pairs(np.concatenate(ms) if ms else np.empty(0, np.int64),
                                    np.concatenate(bs) if bs else np.empty(0, np.int32), np.asarray(po, np.int64), nb, dev,
                                    want_hits=True)
    off2, cnt2 = flat.pair_off.cpu().numpy(), flat.ucount.cpu().numpy()
    ids2, hits2 = flat.cand_ids.cpu().numpy(), flat.hits.cpu().numpy()
    assert np.array_equal(cnt2, cnt)
    for qi in range(nq):
        assert list(zip(ids2[off2[qi]:off2[qi] + cnt2[qi]].tolist(), hits2[off2[qi]:off2[qi] + cnt2[qi]].tolist())) == want[qi]
    # the mirror is reused while the store keeps its segments, rebuilt when one is replaced
    assert mirror.uploads == nseg and mirror.table(segs, dev)[0] is desc and mirror.uploads == nseg
    mirror.table(segs[1:], dev)
    assert mirror.uploads == nseg


def test_large_ids_and_the_limits_of_the_item_layout():
    import torch

    from lshrs_amd import _query_device as qd
    from lshrs_amd.packed_ops import _csr_host

    dev = torch.device("cuda", 0)
    nb, bb = 16, 2
    big = (1 << 59) - 5                                                     # 63 - 4 band bits
    ids = np.array([big, 7, big - 1, 1 << 40], dtype=np.int64)
    keys = np.zeros((4, nb, bb), dtype=np.uint8)
    keys[:, :, 0] = np.arange(nb)[None, :]
    keys[1, 3:, 1] = 1                                                      # id 7 shares three bands only
    seg = _csr_host(ids, keys)
    mirror = qd.DeviceBuckets()                                             # (owns the device arrays the descriptors point to)
    desc, n, max_id = mirror.table([seg], dev)
    lists = qd.candidates_from_index(torch.from_numpy(keys[:1].copy()).to(dev), desc, n, max_id, want_hits=True)
    u = int(lists.ucount.cpu()[0])
    assert lists.cand_ids.cpu().numpy()[:u].tolist() == [1 << 40, big - 1, big, 7]
    assert lists.hits.cpu().numpy()[:u].tolist() == [16, 16, 16, 3]
    seg2 = _csr_host(np.array([1 << 59], dtype=np.int64), keys[:1])
    with pytest.raises(qd.TooLarge):
        d2, n2, m2 = mirror.table([seg2], dev)
        qd.candidates_from_index(torch.from_numpy(keys[:1].copy()).to(dev), d2, n2, m2)


def test_the_cut_is_the_references_ceil():
    """keep = min(max(1, ceil(n * top_p)), top_k) in the double arithmetic of `math.ceil(len(scored) * top_p)`
    (lshrs/core/main.py:652-657), 0 for an empty list; offsets, sum and maximum of the scan."""
    import torch

    from lshrs_amd import _native
    from lshrs_amd import _query_device as qd

    lib = _native.load()
    dev = torch.device("cuda", 0)
    counts = np.r_[np.arange(0, 3000), [16384, 9999, 1, 0, 7]].astype(np.int32)
    cd = torch.from_numpy(counts).to(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for top_k, top_p in [(None, 0.5), (None, 1.0), (10, 0.5), (3, 1.0), (None, 0.3), (None, 1e-9), (None, 0.9999999), (None, 0.1),
                         (None, 1 / 3), (7, 0.7), (10, None), (None, None), (1, None), (None, 0.95), (None, 0.05)]:
        keep = torch.empty(len(counts), dtype=torch.int32, device=dev)
        off, totals = qd._scan(torch, lib, cd, len(counts), dev, stream, top_k=-1 if top_k is None else top_k,
                               top_p=-1.0 if top_p is None else top_p, keep_out=keep)
        want = []
        for n in counts.tolist():
            if n == 0:
                want.append(0)
            elif top_p is None:
                want.append(n if top_k is None else min(n, top_k))
            else:
                lim = max(1, math.ceil(n * top_p))
                want.append(min(min(lim, n), top_k) if top_k is not None else min(lim, n))
        assert keep.cpu().numpy().tolist() == want, (top_k, top_p)
        assert off.cpu().numpy().tolist() == np.r_[0, np.cumsum(want)].tolist()
        assert totals.cpu().numpy().tolist() == [sum(want), max(want)]
    off, totals = qd._scan(torch, lib, cd, len(counts), dev, stream)
    assert off.cpu().numpy().tolist() == np.r_[0, np.cumsum(counts.astype(np.int64))].tolist()
    assert totals.cpu().numpy().tolist() == [int(counts.sum()), 16384]


def _clustered(rng, n, dim, clusters, spread):
    centers = rng.standard_normal((clusters, dim)).astype(np.float32)
    return (np.repeat(centers, n // clusters, axis=0) + spread * rng.standard_normal((n, dim))).astype(np.float32)


def _same_ranking(got, want, tol=1e-5, gap=2e-5):
    """Lists of (id, score): equal lengths; scores within `tol` wherever the ids agree; where they do not, the two are
    near-ties of the reference - scores at most `gap` apart (the reference's own order among ties is unspecified, and a tie
    may straddle the cut: the partner is then not in the list at all)."""
    assert len(got) == len(want)
    for j, ((gi, gs), (wi, ws)) in enumerate(zip(got, want)):
        if gi == wi:
            assert abs(gs - ws) <= tol, (j, gi, gs, ws)
        else:
            assert abs(gs - ws) <= tol + gap, (j, gi, wi, gs, ws)


@pytest.mark.parametrize("dim,num_perm,nb,r,n,clusters,spread", [
    (64, 64, 16, 4, 3000, 300, 0.35),          # one-byte keys, 16 crowded buckets per band: lists of thousands, many equal counts
    (768, 256, 16, 16, 4000, 400, 0.3),        # config 2's shape
    (96, 512, 16, 32, 3000, 300, 0.25),        # four-byte keys (config 5's layout): bisection on 32-bit codes
    (50, 40, 8, 5, 2000, 100, 0.3),            # dim % 4 != 0, five-row bands
])
def test_query_many_on_the_device_equals_the_reference_flow(dim, num_perm, nb, r, n, clusters, spread):
    """>= 1 000 queries per shape through LSHRS.query_many (engine="device") against oracle.query_literal on the same store:
    top_k by collisions (equal counts -> id order, empty buckets, ids in several bands), top_p with its ceil, top_k AND top_p;
    list form, array form, the host engine and - one query at a time - `query` itself."""
    import torch

    from lshrs_amd import LSHRS, InMemoryStorage
    from oracle import lshrs_oracle as O

    rng = np.random.default_rng(dim * 7 + nb)
    data = _clustered(rng, n, dim, clusters, spread)
    store = InMemoryStorage()
    idx = LSHRS(dim=dim, num_perm=num_perm, num_bands=nb, rows_per_band=r, storage=store, packed_ingest=True, seed=42,
                vector_fetch_fn=lambda ids: data[np.asarray(ids)])
    third = n // 3
    idx.index(np.arange(third), data[:third])                        # three calls: three array segments
    idx.index(np.arange(third, 2 * third), data[third:2 * third])
    idx.index(np.arange(2 * third, n), data[2 * third:])
    idx.index(np.arange(100), data[:100])                            # ... and 100 ids indexed twice (the same buckets: sets)
    assert len(store.array_segments((r + 7) // 8)) == 4 and not store.batches
    nq = 1000
    rows = rng.choice(n, nq, replace=False)
    queries = (data[rows] + 0.05 * rng.standard_normal((nq, dim))).astype(np.float32)
    queries[::50] = rng.standard_normal((len(queries[::50]), dim)).astype(np.float32)      # strangers: mostly empty buckets
    P = idx._hasher.projections
    fetch = lambda ids: data[np.asarray(ids)]  # noqa: E731
    corpus = torch.from_numpy(data).cuda()

    lit_all = [O.query_literal(store, P, dim, q, top_k=None) for q in queries]
    assert (r <= 5 or any(len(w) == 0 for w in lit_all)) and any(len(w) > 5 for w in lit_all)
    got_all = idx.query_many(queries, top_k=None, engine="device")
    assert got_all == lit_all
    # the device arrays the ingest grouped these buckets into are the ones the lookup reads (keys of 1 or 2 bytes: DeviceCSRJob):
    # nothing is uploaded for them; wider keys are grouped by a device sort whose result lives on the host: uploaded once
    mirror = idx._dev_buckets
    assert (mirror.kept_from_ingest, mirror.uploads) == ((4, 0) if (r + 7) // 8 <= 2 else (0, 4)), (mirror.kept_from_ingest, mirror.uploads)
    assert idx.query_many(queries, top_k=10, engine="device") == [w[:10] for w in lit_all]
    assert idx.query_many(queries, top_k=1, engine="device") == [w[:1] for w in lit_all]
    assert idx.query_many(queries, top_k=None, engine="host") == lit_all
    ids, scores, bounds = idx.query_many(queries, top_k=7, return_arrays=True)
    assert scores is None and ids.dtype == np.int64 and bounds.shape == (nq + 1,)
    assert [ids[bounds[i]:bounds[i + 1]].tolist() for i in range(nq)] == [w[:7] for w in lit_all]

    # (the literal rerank normalises candidate by candidate in Python: 170 queries per setting, 40 where the lists are thousands long)
    sample = np.r_[0:120, 950:1000] if r > 4 else np.r_[0:30, 990:1000]
    for top_k, top_p in [(None, 0.5), (None, 1.0), (5, 0.3), (3, 1.0), (None, 0.01)]:
        want = [O.query_literal(store, P, dim, queries[i], top_k=top_k, top_p=top_p, fetch=fetch) for i in sample]
        on_corpus = idx.query_many(queries, top_k=top_k, top_p=top_p, corpus=corpus, engine="device")
        fetched = idx.query_many(queries[sample], top_k=top_k, top_p=top_p, engine="device")
        hosted = idx.query_many(queries[sample], top_k=top_k, top_p=top_p, corpus=corpus, engine="host")
        for j, i in enumerate(sample):
            _same_ranking(on_corpus[i], want[j])
            _same_ranking(fetched[j], want[j])
            _same_ranking(hosted[j], want[j])
            n_cand = len(lit_all[i])
            lim = 0 if n_cand == 0 else max(1, math.ceil(n_cand * top_p))
            assert len(on_corpus[i]) == (min(lim, top_k) if top_k is not None else lim)
        ids, scores, bounds = idx.query_many(queries, top_k=top_k, top_p=top_p, corpus=corpus, return_arrays=True)
        assert scores.dtype == np.float32 and len(ids) == len(scores) == bounds[-1]
        assert [list(zip(ids[bounds[i]:bounds[i + 1]].tolist(), scores[bounds[i]:bounds[i + 1]].astype(np.float64).tolist()))
                for i in range(nq)] == on_corpus
    # one query at a time: the reference's own entry points on the attached corpus
    idx.set_corpus(corpus)
    for i in sample[:40]:
        _same_ranking(idx.get_above_p(queries[i], p=0.5),
                      O.query_literal(store, P, dim, queries[i], top_k=None, top_p=0.5, fetch=fetch))
        assert idx.get_top_k(queries[i], topk=4) == lit_all[i][:4]
    # deleting ids replaces segments: the mirror follows
    gone = [int(w[0]) for w in lit_all[:200] if w]
    idx.delete(gone)
    assert idx.query_many(queries[:200], top_k=5, engine="device") == [
        O.query_literal(store, P, dim, q, top_k=5) for q in queries[:200]]
    idx.clear()
    assert idx.query_many(queries[:5], top_k=5, engine="device") == [[]] * 5
    assert idx.query_many(queries[:5], top_k=None, top_p=0.5, engine="device") == [[]] * 5


class _GetBucketOnly:
    """The reference's storage interface and nothing else (lshrs/storage/redis.py: batch_add, get_bucket ...)."""

    def __init__(self):
        from lshrs_amd import InMemoryStorage

        self._inner = InMemoryStorage()

    def batch_add(self, ops):
        self._inner.batch_add(ops)

    def get_bucket(self, band_id, hash_val):
        return self._inner.get_bucket(band_id, hash_val)

    def remove_indices(self, indices):
        self._inner.remove_indices(indices)

    def clear(self):
        self._inner.clear()

    def close(self):
        pass


def test_stores_with_get_bucket_only_count_on_the_device_too():
    """Redis-like stores: one get_bucket per (query, band) as the reference does (main.py:1103), the (member, band) pairs
    uploaded flat, sort / count / order / rerank on the device.  Also a store holding BOTH op-tuple buckets and array segments."""
    import torch

    from lshrs_amd import LSHRS, InMemoryStorage
    from oracle import lshrs_oracle as O

    rng = np.random.default_rng(5)
    dim, n = 128, 3000
    data = _clustered(rng, n, dim, 150, 0.3)
    queries = (data[rng.choice(n, 300, replace=False)] + 0.05 * rng.standard_normal((300, dim))).astype(np.float32)
    corpus = torch.from_numpy(data).cuda()
    fetch = lambda ids: data[np.asarray(ids)]  # noqa: E731
    for store, packed in ((_GetBucketOnly(), "auto"), (InMemoryStorage(), False)):
        idx = LSHRS(dim=dim, num_perm=64, storage=store, packed_ingest=packed, vector_fetch_fn=fetch)
        idx.index(np.arange(n), data)
        if isinstance(store, InMemoryStorage):
            idx._packed_ingest = True
            idx.index(np.arange(n, n + 500), data[:500])                      # tuples AND arrays in one store
            assert store.array_segments(1) is None and store.prefers_batched_lookup
        P = idx._hasher.projections
        want = [O.query_literal(store, P, dim, q, top_k=None) for q in queries]
        assert idx.query_many(queries, top_k=None, engine="device") == want
        assert idx.query_many(queries, top_k=6, engine="device") == [w[:6] for w in want]
        table = corpus if not isinstance(store, InMemoryStorage) else torch.cat([corpus, corpus[:500]])
        full = np.concatenate([data, data[:500]])
        got = idx.query_many(queries[:60], top_k=None, top_p=0.5, corpus=table, engine="device")
        for g, q in zip(got, queries[:60]):
            _same_ranking(g, O.query_literal(store, P, dim, q, top_k=None, top_p=0.5, fetch=lambda ids: full[np.asarray(ids)]))


def test_lists_beyond_the_lds_network_go_through_global_memory():
    """A query whose buckets hold more than LSHRS_QUERY_MAX_PAIRS members (here 24 000 and 320 000 pairs; 1 500 and 20 000 distinct
    candidates - the second also beyond the rank kernel's network): gathered, sorted, counted and ordered through global memory
    (`lshrs_query_collide_big_i64`: K3's long-list network), ranked by `lshrs_topk_desc_f32` - the literal flow's answers, in one
    batch with ordinary queries, through `query_many` and through `get_top_k` / `get_above_p`."""
    import torch

    from lshrs_amd import LSHRS, InMemoryStorage, _native
    from lshrs_amd import _query_device as qd
    from oracle import lshrs_oracle as O

    assert _native.QUERY_MAX_PAIRS == 16384
    rng = np.random.default_rng(9)
    dim = 32
    base = rng.standard_normal(dim).astype(np.float32)
    other = rng.standard_normal((300, dim)).astype(np.float32)
    for n in (1500, 20_000):
        near = (base[None, :] + 1e-4 * rng.standard_normal((n, dim))).astype(np.float32)
        data = np.concatenate([near, other])
        store = InMemoryStorage()
        idx = LSHRS(dim=dim, num_perm=64, storage=store, packed_ingest=True)
        idx.index(np.arange(n), near)
        idx.index(np.arange(n, n + 300), other)
        idx.index(np.arange(50), near[:50])                               # ids indexed twice: once per bucket
        P = idx._hasher.projections
        q = np.stack([base, -base, other[3] + 0.01, base * 2.0, other[7]]).astype(np.float32)
        want = [O.query_literal(store, P, dim, v, top_k=7) for v in q]
        assert len(want[0]) == 7 and idx.query_many(q, top_k=7, engine="device") == want
        assert idx.last_query_stats["longest_list"] == 16384 and idx.last_query_stats["pairs"] >= 2 * 16 * n
        assert idx.query_many(q, top_k=None, engine="device") == [O.query_literal(store, P, dim, v, top_k=None) for v in q]
        assert idx.query_many(q, top_k=7, engine="host") == want
        corpus = torch.from_numpy(data).cuda()
        fetch = lambda ids: data[np.asarray(ids)]  # noqa: E731
        for top_k, top_p in ((None, 0.002), (5, 1.0), (None, 1.0)):
            got = idx.query_many(q, top_k=top_k, top_p=top_p, corpus=corpus, engine="device")
            for g, v in zip(got, q):
                _same_ranking(g, O.query_literal(store, P, dim, v, top_k=top_k, top_p=top_p, fetch=fetch), gap=1e-4)
        # one query per call: the chain says "beyond my capacity", the batch form answers
        assert idx.get_top_k(base, topk=7) == want[0] and idx.get_top_k(-base, topk=7) == want[1]
        idx.set_corpus(corpus)
        _same_ranking(idx.get_above_p(base, p=0.002), O.query_literal(store, P, dim, base, top_k=None, top_p=0.002, fetch=fetch), gap=1e-4)
    # pairs handed over by the host (a store with get_bucket only) are not taken beyond the LDS network: the host counts
    with pytest.raises(qd.TooLarge):
        qd.candidates_from_pairs(np.zeros(20_000, np.int64), np.zeros(20_000, np.int32), np.array([0, 20_000], np.int64), 16,
                                 torch.device("cuda", 0))


def test_errors_are_the_references():
    import torch

    from lshrs_amd import LSHRS, InMemoryStorage

    rng = np.random.default_rng(2)
    data = _clustered(rng, 600, 32, 30, 0.2)
    idx = LSHRS(dim=32, num_perm=16, storage=InMemoryStorage(), packed_ingest=True)
    idx.index(np.arange(600), data)
    q = data[400:408] + 0.01
    with pytest.raises(ValueError, match="Cannot index zero vector"):
        idx.query_many(np.r_[q, np.zeros((1, 32), np.float32)], top_k=3)
    with pytest.raises(RuntimeError, match="vector_fetch_fn must be supplied"):
        idx.query_many(q, top_k=None, top_p=0.5)
    corpus = torch.from_numpy(data).cuda()
    short = corpus[:300]
    with pytest.raises(IndexError, match="out of range"):
        idx.query_many(q, top_k=None, top_p=1.0, corpus=short)
    dead = corpus.clone()
    dead[idx.query_many(q[:1], top_k=1)[0][0]] = 0
    with pytest.raises(ValueError, match="Cannot normalize zero vector"):
        idx.query_many(q[:1], top_k=None, top_p=1.0, corpus=dead)
    with pytest.raises(ValueError, match="engine must be"):
        idx.query_many(q, engine="gpu")
    ids, scores, bounds = idx.query_many(np.empty((0, 32), np.float32), top_p=0.5, corpus=corpus, return_arrays=True)
    assert len(ids) == 0 and len(scores) == 0 and bounds.tolist() == [0]


def test_one_query_is_one_chain_of_launches_and_equals_the_reference_flow():
    """`get_top_k` / `get_above_p` / `query` - the reference's own calling pattern (main.py:524-658) - through
    `_query_device.OneQuery`: signature kernel, lookup, collide, cut (and the rerank on the attached corpus) enqueued back to
    back with fixed capacities, ONE wait, the answer in pinned memory.  Against oracle.query_literal: 400 queries (strangers with
    empty buckets among them), every argument form, the reference's error order, concurrent callers."""
    import threading

    import torch

    from lshrs_amd import LSHRS, InMemoryStorage
    from oracle import lshrs_oracle as O

    rng = np.random.default_rng(21)
    dim, n = 768, 5000
    data = _clustered(rng, n, dim, 250, 0.3)
    store = InMemoryStorage()
    idx = LSHRS(dim=dim, num_perm=256, storage=store, packed_ingest=True)
    idx.index(np.arange(2500), data[:2500])
    idx.index(np.arange(2500, n), data[2500:])
    P = idx._hasher.projections
    fetch = lambda ids: data[np.asarray(ids)]  # noqa: E731
    queries = (data[rng.choice(n, 400, replace=False)] + 0.05 * rng.standard_normal((400, dim))).astype(np.float32)
    queries[::40] = rng.standard_normal((10, dim)).astype(np.float32)
    calls = []
    run = idx._query_one_device
    idx._query_one_device = lambda *a: (calls.append(1), run(*a))[1]
    got = [idx.get_top_k(q, topk=5) for q in queries]
    assert calls and idx._one_query, "the single-query chain was not taken"
    want = [O.query_literal(store, P, dim, q, top_k=5) for q in queries]
    assert got == want and any(len(w) == 0 for w in want)
    assert [idx.query(q, top_k=None) for q in queries[:100]] == [O.query_literal(store, P, dim, q, top_k=None) for q in queries[:100]]
    assert all(type(i) is int for r in got for i in r)
    # rerank on the attached corpus
    idx.set_corpus(torch.from_numpy(data).cuda())
    for q in queries[:120]:
        for kw in ({"top_k": None, "top_p": 0.5}, {"top_k": 3, "top_p": 1.0}, {"top_k": None, "top_p": 0.01}):
            _same_ranking(idx.query(q, **kw), O.query_literal(store, P, dim, q, fetch=fetch, **kw))
    res = idx.get_above_p(queries[1], p=0.5)
    assert all(type(i) is int and type(s) is float for i, s in res)
    # errors, in the reference's order: a zero vector first; the arguments only when there are candidates
    with pytest.raises(ValueError, match="Cannot index zero vector"):
        idx.get_top_k(np.zeros(dim, np.float32), topk=0)
    with pytest.raises(ValueError, match="top_k must be greater than zero"):
        idx.get_top_k(queries[1], topk=0)
    with pytest.raises(ValueError, match=r"top_p must be within the range \(0, 1\]"):
        idx.get_above_p(queries[1], p=1.5)
    with pytest.raises(ValueError, match="top_k must be greater than zero"):
        idx.query(queries[1], top_k=-2, top_p=0.5)
    stranger = next(q for q, w in zip(queries, want) if not w)
    assert idx.get_top_k(stranger, topk=0) == [] and idx.get_above_p(stranger, p=7.0) == []     # (no candidates: nothing is checked)
    # concurrent callers (the reference's test_concurrency pattern): one chain at a time, every answer right
    out = [None] * 64
    def work(j):
        out[j] = idx.get_top_k(queries[j], topk=5)
    threads = [threading.Thread(target=work, args=(j,)) for j in range(64)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert out == want[:64]
    # a store that keeps tuples, a rerank through vector_fetch_fn: the host-counted path answers (same lists)
    n_calls = len(calls)
    plain = LSHRS(dim=dim, num_perm=256, storage=InMemoryStorage(), packed_ingest=False, vector_fetch_fn=fetch)
    plain.index(np.arange(600), data[:600])
    assert plain.get_top_k(data[5] + 0.01, topk=3) == O.query_literal(plain._storage, P, dim, data[5] + 0.01, top_k=3)
    _same_ranking(plain.get_above_p(data[5] + 0.01, p=0.5), O.query_literal(plain._storage, P, dim, data[5] + 0.01, top_k=None, top_p=0.5, fetch=fetch))
    assert not plain._one_query and len(calls) == n_calls


def test_lists_of_every_length_through_the_register_and_the_lds_sorts():
    """Round 6: lists of up to 256 / 512 items are sorted in registers (lane shuffles + two trips through LDS), longer ones
    through LDS - every length around the seams, with repeated (member, band) pairs and members in many bands: the candidates
    and their collision counts against a dict, and the ranked answer (scores descending, a permutation of the candidates,
    every score the cosine) for the same lists."""
    import torch

    from lshrs_amd import _query_device as qd

    rng = np.random.default_rng(99)
    nb, dim, m = 16, 32, 6000
    lengths = [1, 2, 3, 5, 31, 63, 64, 65, 127, 128, 129, 200, 255, 256, 257, 300, 400, 511, 512, 513, 700, 1023, 1024, 1025, 3000, 9000]
    ms, bs, po, want = [], [], [0], []
    for L in lengths:
        pool = rng.choice(m, size=max(1, L // 3), replace=False)             # few distinct members: real collisions
        mem = rng.choice(pool, size=L).astype(np.int64)
        band = rng.integers(0, nb, size=L).astype(np.int32)
        ms.append(mem)
        bs.append(band)
        po.append(po[-1] + L)
        counts = {}
        for mm, bb in set(zip(mem.tolist(), band.tolist())):                 # buckets are sets: a pair counts once
            counts[mm] = counts.get(mm, 0) + 1
        want.append(sorted(counts.items(), key=lambda kv: (-kv[1], kv[0])))
    ms.append(np.full(300, 7, np.int64))                                     # one member, one band, 300 times: one candidate, one hit
    bs.append(np.full(300, 3, np.int32))
    po.append(po[-1] + 300)
    want.append([(7, 1)])
    dev = torch.device("cuda", 0)
    lists = qd.candidates_from_pairs(np.concatenate(ms), np.concatenate(bs), np.asarray(po, np.int64), nb, dev, want_hits=True)
    off, cnt = lists.pair_off.cpu().numpy(), lists.ucount.cpu().numpy()
    ids, hits = lists.cand_ids.cpu().numpy(), lists.hits.cpu().numpy()
    for qi, w in enumerate(want):
        got = list(zip(ids[off[qi]:off[qi] + cnt[qi]].tolist(), hits[off[qi]:off[qi] + cnt[qi]].tolist()))
        assert got == w, (qi, len(w))
    corpus = rng.standard_normal((m, dim)).astype(np.float32)
    queries = rng.standard_normal((len(want), dim)).astype(np.float32)
    r_ids, r_scores, bounds = qd.rank_and_cut(lists, None, 1.0, queries_dev=torch.from_numpy(queries).to(dev),
                                              corpus=torch.from_numpy(corpus).to(dev))
    cn = corpus / np.linalg.norm(corpus, axis=1, keepdims=True)
    for qi, w in enumerate(want):
        g_ids, g_sc = r_ids[bounds[qi]:bounds[qi + 1]], r_scores[bounds[qi]:bounds[qi + 1]]
        assert sorted(g_ids.tolist()) == sorted(i for i, _ in w), qi
        assert np.all(g_sc[:-1] >= g_sc[1:]), qi
        ref = cn[g_ids] @ (queries[qi] / np.linalg.norm(queries[qi]))
        assert np.abs(ref - g_sc).max() <= 1e-5, qi


def test_queries_that_live_on_the_gpu_are_taken_as_they_are():
    """Round 6: `query_many` with a torch tensor on the GPU - contiguous, a strided view, float64 - gives what the same rows
    give as a NumPy array, in both result forms and with the rerank; a wrong shape is the reference's error."""
    import torch

    from lshrs_amd import LSHRS, InMemoryStorage

    rng = np.random.default_rng(21)
    n, dim = 20_000, 64
    data = rng.standard_normal((n, dim)).astype(np.float32)
    idx = LSHRS(dim=dim, num_perm=64, storage=InMemoryStorage(), packed_ingest=True)
    idx.index(np.arange(n), data)
    dev = torch.device("cuda", 0)
    idx.set_corpus(torch.from_numpy(data).to(dev))
    q = data[rng.choice(n, 500, replace=False)] + 0.05 * rng.standard_normal((500, dim)).astype(np.float32)
    want_k = idx.query_many(q, top_k=7)
    want_p = idx.query_many(q, top_k=None, top_p=0.5, return_arrays=True)
    qd_ = torch.from_numpy(q).to(dev)
    assert idx.query_many(qd_, top_k=7) == want_k
    got_p = idx.query_many(qd_, top_k=None, top_p=0.5, return_arrays=True)
    assert all(np.array_equal(a, b) for a, b in zip(got_p, want_p))
    wide = torch.zeros(500, 2 * dim, device=dev)
    wide[:, :dim] = qd_
    assert idx.query_many(wide[:, :dim], top_k=7) == want_k                          # rows of a wider matrix
    every_other = torch.zeros(1000, dim, device=dev)
    every_other[::2] = qd_
    assert idx.query_many(every_other[::2], top_k=7) == want_k                       # every other row
    assert idx.query_many(qd_.double(), top_k=7) == want_k
    assert idx.query_many(qd_[:0], top_k=7) == []
    with pytest.raises(ValueError, match="Vectors must have shape"):
        idx.query_many(qd_[:, :10], top_k=7)
