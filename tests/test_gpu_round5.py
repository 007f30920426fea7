"""GPU tests of round 5.  Every call goes through the C ABI; the checker is the oracle's literal restatement of
lshrs/hash/lsh.py:96-211."""

from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def _hasher(seed, nb, r, dim, **kw):
    from lshrs_amd import LSHHasher

    return LSHHasher(num_bands=nb, rows_per_band=r, dim=dim, seed=seed, **kw)


# ----------------------------------------------------------------------------- one pass as a pipeline
@pytest.mark.parametrize("nb,dim,n", [(64, 100, 6_000), (16, 33, 5_000), (40, 31, 5_000), (24, 7, 4_000), (8, 2, 3_000),
                                      (128, 770, 3_000), (200, 96, 4_000), (32, 1000, 3_000)])
def test_bands_of_one_row_at_lengths_with_a_tail(torch_mod, nb, dim, n):
    """`rows_per_band = 1`: the host calls sdot - the build's SIMD kernel over the whole 32-element steps, the f32 products of the
    elements behind them summed in a DOUBLE, one rounding (round 5, `tb_model_sdot`).  The plain-load replay follows that at
    every length, true ties included, the reference-literal loop's bytes - behind the exact-f32 kernel (`f32+replay`: round 5) and,
    since round 6, as stage 2 of the split pass wherever its stage 1 takes the shape (`split+replay`: the matrix cores first)."""
    torch = torch_mod
    from oracle.lshrs_oracle import hash_batch_literal_packed

    h = _hasher(19, nb, 1, dim)
    if not h._replay_model():
        pytest.skip("the host BLAS's sdot order is not one the replay knows on this box")
    x = np.random.default_rng(dim).standard_normal((n, dim)).astype(np.float32)
    stack = np.concatenate([np.asarray(p, dtype=np.float64) for p in h.projections])
    special = np.arange(0, n, 20)
    k = min(3, max(1, dim - 1))
    for i in special:                                      # true ties against up to three hyperplanes each
        pl = stack[[(i + t) % nb for t in (0, 7, 11)][:k]]
        v = x[i].astype(np.float64)
        x[i] = (v - (v @ np.linalg.pinv(pl)) @ pl).astype(np.float32)
    got = h.hash_device(torch.from_numpy(x).cuda())
    st = dict(h.last_stats)
    assert st["route"] in ("f32+replay", "split+replay") and st["tie_break_engine"] == "device-replay", st
    assert st["tie_pairs"] >= special.size // 2, st
    if st["route"] == "split+replay":           # stage 2 measured stage 1 against the host's sdot on every flagged projection, and audited the rest
        assert st["flagged"] >= special.size and st["audit_sign_disagreements"] == 0 and st["audited_unflagged"] > 0, st
        assert 0.0 < st["max_dev_units"] <= h.window_info["window_units_worst_case_row"], st
    want = hash_batch_literal_packed(h.projections, x)
    assert np.array_equal(got.cpu().numpy(), want), int((got.cpu().numpy() != want).any(axis=(1, 2)).sum())
    f32 = _hasher(19, nb, 1, dim, precision="f32")       # the other route: the same bytes
    assert torch.equal(f32.hash_device(torch.from_numpy(x).cuda()), got) and f32.last_stats["route"] == "f32+replay"
    assert np.array_equal(h.hash_batch_packed(x[:60]), want[:60])
    assert h.hash_vector(x[special[2]]).as_tuple() == tuple(bytes(kk) for kk in want[special[2]])


# ----------------------------------------------------------------------------- keys pinned to a named BLAS build
def test_named_reference_blas_keeps_the_device_route_whatever_the_host_blas(torch_mod, monkeypatch):
    """`reference_blas="openblas-skylakex"` with the licence check answering "not recognised": route `split+replay`, no host
    engine; for dim % 4 == 0 both named builds give the same keys, and where this host IS an OpenBLAS of that kind, the
    reference-literal loop's bytes."""
    import time

    from lshrs_amd import _hostblas
    from oracle.lshrs_oracle import hash_batch_literal_packed

    torch = torch_mod
    base = _hasher(42, 16, 16, 768)
    host_model = base._replay_model()                      # (the real answer of this host, before the patch)
    monkeypatch.setattr(_hostblas, "blas_order_model", lambda planes: 0)
    n = 400_000
    x = torch.randn(n, 768, device="cuda", generator=torch.Generator("cuda").manual_seed(21))
    lost = _hasher(42, 16, 16, 768)
    assert lost._replay_model() == 0 and lost._route(n, "host", aligned=True, short_stride=True, host_rows=False)[0] != "split+replay"
    sk = _hasher(42, 16, 16, 768, reference_blas="openblas-skylakex")
    hw = _hasher(42, 16, 16, 768, reference_blas="openblas-haswell")
    ks = sk.hash_device(x)
    st = dict(sk.last_stats)
    assert st["route"] == "split+replay" and st["tie_break_engine"] == "device-replay" and st["reference_blas"] == "openblas-skylakex"
    assert st["audit_sign_disagreements"] == 0 and "audited" not in st        # (no live audit against a NumPy that is another BLAS)
    assert torch.equal(hw.hash_device(x), ks)
    t0 = time.perf_counter()
    for _ in range(20):
        sk.hash_device(x)
    torch.cuda.synchronize()
    print(f"pinned model: {n * 20 / (time.perf_counter() - t0) / 1e6:.0f} M vec/s on 400 000 x 768 (bench.py pinned_model_mode has the 1 M figure)")
    if host_model == 1:
        rows = np.r_[0:1500, n - 1500:n]
        assert np.array_equal(ks[rows].cpu().numpy(), hash_batch_literal_packed(base.projections, x[rows].cpu().numpy()))
        monkeypatch.undo()
        assert torch.equal(_hasher(42, 16, 16, 768).hash_device(x), ks)


@pytest.mark.parametrize("nb,r,dim", [(16, 16, 102), (8, 5, 99), (64, 1, 100), (16, 1, 768), (16, 16, 301), (20, 10, 767)])
def test_named_builds_differ_exactly_where_their_libraries_do(torch_mod, nb, r, dim):
    """dim % 4 != 0 (the scalar tail) and one-row bands (sdot): the two named builds sum differently - on rows cancelled against a
    hyperplane the keys of each are the signs of ITS model (`lshrs_tb_model_row_dot`), and the one that is this host's gives the
    reference-literal loop's bytes."""
    from lshrs_amd import _hostblas
    from oracle.lshrs_oracle import hash_batch_literal_packed

    torch = torch_mod
    n = 4_000
    lib = _hostblas.load()
    x = np.random.default_rng(dim).standard_normal((n, dim)).astype(np.float32)
    base = _hasher(23, nb, r, dim)
    stack = np.concatenate([np.asarray(p, dtype=np.float64) for p in base.projections])
    targets = []
    for i in range(0, n, 10):                              # a true tie against one hyperplane per salted row
        j = (7 * i) % (nb * r)
        v = x[i].astype(np.float64)
        x[i] = (v - (v @ stack[j]) / (stack[j] @ stack[j]) * stack[j]).astype(np.float32)
        targets.append((i, j))
    xd = torch.from_numpy(x).cuda()
    keys = {}
    for build in ("openblas-skylakex", "openblas-haswell"):
        h = _hasher(23, nb, r, dim, reference_blas=build)
        keys[build] = h.hash_device(xd).cpu().numpy()
        assert h.last_stats["route"] == ("split+replay" if h._split_applies(n, replay=True) else "f32+replay"), h.last_stats
        model = h._replay_model()
        for i, j in targets:                               # the tied projection's key bit is the sign of THAT build's value
            b, bit = j // r, j % r
            p32 = np.ascontiguousarray(base.projections[b][bit], dtype=np.float32)
            y = lib.lshrs_tb_model_row_dot(p32.ctypes.data, x[i].ctypes.data, dim, model, bit, r)
            assert ((keys[build][i, b, bit >> 3] >> (bit & 7)) & 1) == int(y > 0), (build, i, j, y)
    host_model = base._replay_model()
    if host_model in (1, 2):
        mine = "openblas-skylakex" if host_model == 1 else "openblas-haswell"
        assert np.array_equal(keys[mine], hash_batch_literal_packed(base.projections, x))
    differ = int((keys["openblas-skylakex"] != keys["openblas-haswell"]).any(axis=(1, 2)).sum())
    print(f"{nb} x {r} x {dim}: rows whose keys differ between the two builds: {differ} of {len(targets)} salted")


# ----------------------------------------------------------------------------- the pipelined ingest
def _store_dict(store):
    return {k: set(v) for k, v in store.bucket_contents().items()}


def _tuple_store(idx_kw, ids, x, upto):
    """What the reference's per-vector flow leaves in the storage for rows [0, upto): the op-tuple path of this build (itself
    pinned to the reference's recorded batches by g5 and test_random_api_sequences_equal_the_literal_flow)."""
    from lshrs_amd import LSHRS, InMemoryStorage

    store = InMemoryStorage()
    idx = LSHRS(storage=store, packed_ingest=False, **idx_kw)
    if upto:
        idx.index(list(ids[:upto]), x[:upto])
    return _store_dict(store)


@pytest.mark.parametrize("nb,r,dim,n", [(16, 16, 768, 70_000), (16, 4, 128, 90_000), (16, 32, 256, 40_000)])
def test_pipelined_index_leaves_the_buckets_of_the_per_vector_flow(torch_mod, nb, r, dim, n):
    """`LSHRS.index` on a storage that takes bucket arrays: chunks are copied, hashed, grouped on the device and stored while
    the next chunk is on the link (lshrs_amd/_ingest.py).  Same buckets as the op-tuple path; a zero vector or a negative id -
    inside a chunk, on a chunk boundary, in the first row - raises the reference's error with exactly the rows in front stored."""
    from lshrs_amd import LSHRS, InMemoryStorage, _ingest

    kw = dict(dim=dim, num_perm=nb * r, num_bands=nb, rows_per_band=r, seed=9)
    x = np.random.default_rng(n).standard_normal((n, dim)).astype(np.float32)
    ids = np.random.default_rng(1).permutation(10 * n)[:n].astype(np.int64)
    old = _ingest.CsrIngest.chunk_rows
    _ingest.CsrIngest.chunk_rows = 16_384            # several chunks per call (the stream's smallest)
    try:
        store = InMemoryStorage()
        idx = LSHRS(storage=store, **kw)
        idx.index(ids, x)
        assert _store_dict(store) == _tuple_store(kw, ids, x, n)
        assert len(store.packed_batches) >= (n // 16_384 if nb * ((r + 7) // 8) * 8 >= 128 or dim <= 256 else 1) or len(store.packed_batches) >= 1
        assert sum(v for v, _ in store.packed_batches) == n and not store.batches
        for bad_row, kind in ((40_000 if n > 40_000 else 20_000, "zero"), (16_384, "zero"), (0, "zero"), (33_333, "neg"), (16_384 * 2, "neg")):
            xb, ib = x.copy(), ids.copy()
            if kind == "zero":
                xb[bad_row] = 0.0
                xb[bad_row + 5] = 0.0                      # (a later one must not matter)
            else:
                ib[bad_row] = -7
                xb[bad_row + 9] = 0.0                      # (a zero vector BEHIND the negative id: the id's error wins)
            s2 = InMemoryStorage()
            with pytest.raises(ValueError, match="zero vector" if kind == "zero" else "non-negative"):
                LSHRS(storage=s2, **kw).index(ib, xb)
            assert _store_dict(s2) == _tuple_store(kw, ib, xb, bad_row), (bad_row, kind)
    finally:
        _ingest.CsrIngest.chunk_rows = old


def test_eight_lanes_deal_loader_batches_round_robin_and_store_them_in_order(torch_mod):
    """`LSHRS(devices=[0] * 8)` (SURVEY §8e without the hardware: eight lanes on the one GPU): `create_signatures` deals whole
    loader batches - ragged ones - to the lanes round-robin; the storage receives their bucket arrays in batch order, the same
    sequence a single device produces.  A bad row in batch 11 stops the ingest there; a loader that raises is the caller's error."""
    from lshrs_amd import LSHRS, InMemoryStorage

    dim = 256
    kw = dict(dim=dim, num_perm=128, num_bands=16, rows_per_band=8, seed=4)
    rng = np.random.default_rng(77)
    sizes = [int(v) for v in rng.integers(600, 9_000, 20)]
    data = [rng.standard_normal((m, dim)).astype(np.float32) for m in sizes]
    starts = np.cumsum([0] + sizes)
    batches = [(list(range(int(starts[i]), int(starts[i + 1]))), data[i]) for i in range(len(sizes))]
    one, eight = InMemoryStorage(), InMemoryStorage()
    LSHRS(storage=one, **kw).create_signatures(format="batches", batches=iter(batches))
    idx8 = LSHRS(storage=eight, devices=[0] * 8, **kw)
    assert len(idx8._ingest_hashers()) == 8
    idx8.create_signatures(format="batches", batches=iter(batches))
    assert eight.packed_batches == one.packed_batches and [v for v, _ in one.packed_batches] == sizes      # same sequence
    assert _store_dict(eight) == _store_dict(one)
    # a zero vector in batch 11, row 17: batches 0..10 whole, 17 rows of batch 11, nothing behind - on eight lanes as on one
    data_bad = [d.copy() for d in data]
    data_bad[11][17] = 0.0
    bad = [(b[0], data_bad[i]) for i, b in enumerate(batches)]
    stores = []
    for devs in (None, [0] * 8):
        st = InMemoryStorage()
        with pytest.raises(ValueError, match="zero vector"):
            LSHRS(storage=st, devices=devs, **kw).create_signatures(format="batches", batches=iter(bad))
        stores.append(st)
    want = InMemoryStorage()
    ref = LSHRS(storage=want, **kw)
    for i in range(11):
        ref.index(batches[i][0], data[i])
    ref.index(batches[11][0][:17], data[11][:17])
    assert _store_dict(stores[0]) == _store_dict(stores[1]) == _store_dict(want)
    assert [v for v, _ in stores[1].packed_batches] == sizes[:11] + [17]

    def raising():
        for i, b in enumerate(batches):
            if i == 6:
                raise RuntimeError("the loader lost its connection")
            yield b

    st = InMemoryStorage()
    with pytest.raises(RuntimeError, match="lost its connection"):
        LSHRS(storage=st, devices=[0] * 8, **kw).create_signatures(format="batches", batches=raising())
    assert [v for v, _ in st.packed_batches] == sizes[:6]
    # one index() call of many rows on several lanes: contiguous slices, same buckets
    big = np.concatenate(data)
    ids = np.arange(big.shape[0], dtype=np.int64)
    a, b = InMemoryStorage(), InMemoryStorage()
    i2 = LSHRS(storage=a, devices=[0, 0], **kw)
    i2.lane_rows = 20_000
    i2.index(ids, big)
    LSHRS(storage=b, **kw).index(ids, big)
    assert _store_dict(a) == _store_dict(b) and len(a.packed_batches) == -(-big.shape[0] // 20_000)


# ----------------------------------------------------------------------------- stage 2 column by column
@pytest.mark.parametrize("nb,r,dim,n,seed", [(16, 32, 1536, 120_000, 7), (16, 16, 768, 90_000, 42), (20, 10, 768, 60_000, 3),
                                              (16, 16, 300, 50_000, 5), (25, 8, 1000, 40_000, 6), (8, 25, 4100 - 4, 9_000, 8),
                                              (8, 7, 200, 30_000, 9), (16, 16, 767, 30_000, 10), (20, 10, 333, 20_000, 11)])   # (+ a scalar tail)
def test_stage2_on_a_column_sorted_list_decides_the_same_bits(torch_mod, nb, r, dim, n, seed):
    """ABI 6, `lshrs_sig_sort`: the flagged projections by key column (stage 1 appends them to their column's segment), stage 2 with
    ONE hyperplane per group of eight (fetched once into LDS).  Same keys as the plain stage 2 - bands of any height, partial k-tiles, 8 m + 4 elements, blocks of
    4096 -, same statistics, the audit sample still verified; and the reference-literal loop's bytes on rows with true ties."""
    from oracle.lshrs_oracle import hash_batch_literal_packed

    torch = torch_mod
    h = _hasher(seed, nb, r, dim)
    if not h._replay_model():
        pytest.skip("the host BLAS's summation order is not one the replay knows on this box")
    x = np.random.default_rng(dim + nb).standard_normal((n, dim)).astype(np.float32)
    stack = np.concatenate([np.asarray(p, dtype=np.float64) for p in h.projections])
    special = np.arange(0, n, 50)
    for i in special:                                      # true ties against three hyperplanes (first, middle, last rows of bands)
        pl = stack[[(i * 7 + t) % (nb * r) for t in (0, r // 2, r - 1)]]
        v = x[i].astype(np.float64)
        x[i] = (v - (v @ np.linalg.pinv(pl)) @ pl).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    h.stage2_sorted = False
    ref = h.hash_device(xd).clone()
    plain = dict(h.last_stats)
    assert plain["route"] == "split+replay"
    for mode in ("buckets", "auto", "buckets"):     # (a second launch reuses the scratch; the launches alternate between two sets of counters)
        h.stage2_sorted = mode
        assert h._stage2_mode() == (None if h._resident_shape() else 1)
        got = h.hash_device(xd)
        st = dict(h.last_stats)
        assert torch.equal(got, ref), mode
        assert (st["flagged"], st["sign_flips"], st["tie_pairs"]) == (plain["flagged"], plain["sign_flips"], plain["tie_pairs"]), (mode, st, plain)
        assert abs(st["max_dev_units"] - plain["max_dev_units"]) <= 1e-3 * max(1.0, plain["max_dev_units"])
        assert st["audited_unflagged"] > 0 and st["audit_sign_disagreements"] == 0, (mode, st)
        # (the salted rows tie against a handful of hyperplanes: those columns may outgrow their segments once - the pass is
        #  repeated with room and the hasher remembers)
        assert st["relaunches"] == 0 or h._bucket_cap_hint > 0, (mode, st)
        if st["relaunches"]:
            h.hash_device(xd)
            assert h.last_stats["relaunches"] == 0
    rows = np.r_[special[:200], n - 300:n]
    assert np.array_equal(ref[rows].cpu().numpy(), hash_batch_literal_packed(h.projections, x[rows]))
    # a list that outgrows its capacity is noticed with the sorted stage 2 as without (rows flagged wholesale), and repeated
    xd[1000:1400] *= 2.0 ** 40
    for mode in ("buckets",):
        h.stage2_sorted = mode
        h._flag_cap_hint = h._bucket_cap_hint = 0
        h._sort_res.clear()
        got = h.hash_device(xd)
        assert h.last_stats["flagged"] >= 400 * nb * r, mode
        h.stage2_sorted = False
        assert torch.equal(got, h.hash_device(xd)), mode
    # rows aligned with ONE hyperplane: that column's segment overflows although the list as a whole would not - noticed, repeated
    p0 = torch.from_numpy(np.asarray(h.projections[0][0], dtype=np.float32)).cuda()
    xa = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    xa[: n // 2] -= ((xa[: n // 2] @ p0) / (p0 @ p0))[:, None] * p0[None, :]
    h.stage2_sorted = "buckets"
    h._flag_cap_hint = h._bucket_cap_hint = 0
    h._sort_res.clear()
    got = h.hash_device(xa)
    assert h.last_stats["flagged"] >= n // 2, h.last_stats
    if h._stage2_mode() == 1:                 # (a resident-image shape keeps its one list: nothing to outgrow)
        assert h.last_stats["relaunches"] >= 1, h.last_stats
    h.stage2_sorted = False
    assert torch.equal(got, h.hash_device(xa))
    assert _hasher(seed, nb, r, dim).stage2_sorted == "auto"


@pytest.mark.parametrize("nb,r,dim,n", [(8, 16, 768, 90_000), (12, 16, 1024, 40_000), (24, 16, 768, 30_000)])
def test_buckets_have_no_segment_for_the_padding_columns(torch_mod, nb, r, dim, n):
    """Fewer key columns than the 256 of a stage-1 column block: a NaN / Inf row makes stage 1 list the zero-padded columns
    behind the last key column too (0 * NaN).  The plain list carries them to stage 2, which skips them; a bucket launch must
    drop them in stage 1 - there is no segment behind the last key column's (round 5: they were written past the scratch)."""
    torch = torch_mod
    h = _hasher(11, nb, r, dim)
    if not h._replay_model():
        pytest.skip("the host BLAS's summation order is not one the replay knows on this box")
    x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(16))
    x[7] = 0.0
    x[9, 3] = float("nan")
    x[n - 5, 0] = float("inf")
    x[1000:1040, 5] = float("nan")
    h.stage2_sorted = False
    ref = h.hash_device(x).clone()
    plain = dict(h.last_stats)
    assert plain["route"] == "split+replay"
    h.stage2_sorted = "buckets"
    assert h._stage2_mode() == 1
    pad = (-(-nb * r // 256) * 256 - nb * r) * 42                # 42 rows flagged wholesale x the padding columns
    for _ in range(3):
        got = h.hash_device(x)
        st = dict(h.last_stats)
        assert torch.equal(got, ref)
        assert st["audited_unflagged"] > 0 and st["audit_sign_disagreements"] == 0 and st["audit_max_window_ratio"] <= 1.0, st
        assert st["flagged"] == plain["flagged"] - pad, (st, plain, pad)


# ----------------------------------------------------------------------------- rows with a scalar tail through the split pass
@pytest.mark.parametrize("nb,r,dim,n,seed", [(16, 16, 102, 60_001, 1), (20, 6, 127, 50_000, 2), (32, 8, 9, 40_000, 3),
                                              (8, 7, 201, 30_000, 4), (16, 4, 33, 70_000, 5), (12, 16, 61, 20_003, 6),
                                              # ... and through sig16_kernel<., PARTIAL>: one element into the last k-tile, three
                                              # short of a whole one, compact column blocks, the zero-padded 256-column image
                                              (16, 16, 301, 30_001, 7), (16, 16, 769, 20_000, 8), (16, 16, 767, 20_000, 9),
                                              (20, 10, 333, 20_000, 10), (8, 16, 771, 20_000, 11), (16, 32, 1025, 12_000, 12)])
def test_rows_with_a_scalar_tail_take_the_split_pass(torch_mod, nb, r, dim, n, seed):
    """dim % 4 != 0 (16 x 16 x 102 of `other_shapes`): stage 1 fetches the row's last dim % 4 elements with its last four and
    shifts them into place (both kernels), stage 2 is the plain-load replay (scalar tail as the host's build compiles it) with
    the margin statistics and the audit sample.  Keys == the exact-f32 route's on every row, == the literal loop's on rows with
    true ties; a NaN row, a zero row, the batch's last row, and rows that are a view into a wider matrix."""
    from oracle.lshrs_oracle import hash_batch_literal_packed

    torch = torch_mod
    h = _hasher(seed, nb, r, dim)
    if not h._replay_model():
        pytest.skip("the host BLAS's summation order is not one the replay knows on this box")
    x = np.random.default_rng(dim + nb).standard_normal((n, dim)).astype(np.float32)
    stack = np.concatenate([np.asarray(p, dtype=np.float64) for p in h.projections])
    special = np.arange(0, n, 40)
    for i in special:                                      # true ties against up to three hyperplanes
        pl = stack[sorted({(i // 40 * 7 + t) % (nb * r) for t in (0, r // 2, r - 1)})][: max(1, min(3, dim - 2))]   # (spread over the columns)
        v = x[i].astype(np.float64)
        x[i] = (v - (v @ np.linalg.pinv(pl)) @ pl).astype(np.float32)
    x[7] = 0.0
    x[9, dim - 1] = float("nan")                           # in the tail
    x[n - 1, dim - 2:] = 3.0
    xd = torch.from_numpy(x).cuda()
    flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
    got = h.hash_device(xd, row_flags=flags)
    st = dict(h.last_stats)
    assert st["route"] == "split+replay" and st["relaunches"] == 0, st
    assert st["audited_unflagged"] > 0 and st["audit_sign_disagreements"] == 0 and st["audit_max_window_ratio"] <= 1.0, st
    assert 0.0 < st["max_dev_units"] < st["tau1_ulps"], st
    assert flags[7].item() == 1 and flags[9].item() == 2 and int(flags.sum()) == 3
    h32 = _hasher(seed, nb, r, dim, precision="f32")
    ref = h32.hash_device(xd)
    assert h32.last_stats["route"] == "f32+replay"
    assert torch.equal(got, ref), f"{int((got != ref).any(dim=2).any(dim=1).sum())} rows differ"
    rows = np.unique(np.concatenate([special, [7, 9, n - 1], np.arange(n - 300, n)]))
    with np.errstate(invalid="ignore"):
        lit = hash_batch_literal_packed(h.projections, x[rows])
    assert np.array_equal(got.cpu().numpy()[rows], lit)
    # a view into a wider matrix: rows at 4-byte addresses, the stride no multiple of four
    wide = torch.zeros(4096, dim + 7, device="cuda")
    wide[:, 3:3 + dim] = xd[:4096]
    wide[:, :3] = float("nan")                             # what lies beside a row must never be read into it
    wide[:, 3 + dim:] = float("inf")
    view = wide[:, 3:3 + dim]
    got_v = h.hash_device(view)
    assert h.last_stats["route"] == "split+replay"
    assert torch.equal(got_v, got[:4096])
    pend = [h.hash_device_async(view), h.hash_device_async(xd)]       # the asynchronous form takes the same rows
    assert torch.equal(pend[0].result(), got_v) and torch.equal(pend[1].result(), got)


def test_aligned_resident_shapes_take_views_at_any_address(torch_mod):
    """dim % 4 == 0 at a resident-image shape, but the rows are a view at a 4-byte address (round 4 sent those to the f32 kernel)."""
    torch = torch_mod
    h = _hasher(3, 16, 4, 128)
    if not h._replay_model():
        pytest.skip("the host BLAS's summation order is not one the replay knows on this box")
    n = 50_000
    wide = torch.randn(n, 131, device="cuda", generator=torch.Generator("cuda").manual_seed(4))
    view = wide[:, 1:129]
    got = h.hash_device(view)
    assert h.last_stats["route"] == "split+replay"
    ref = h.hash_device(view.contiguous())
    assert h.last_stats["route"] == "split+replay" and torch.equal(got, ref)


# ----------------------------------------------------------------------------- fewer than 9 elements
@pytest.mark.parametrize("build", ["openblas-haswell", "openblas-skylakex"])
@pytest.mark.parametrize("nb,r,dim", [(8, 4, 2), (8, 5, 3), (6, 7, 5), (4, 6, 6), (5, 3, 7), (16, 16, 4), (7, 2, 1), (3, 9, 7), (4, 13, 8),
                                      (2, 16, 2), (3, 17, 3), (2, 24, 6), (2, 33, 7), (3, 20, 8), (6, 2, 8), (5, 3, 8)])
def test_fewer_than_nine_elements_on_either_build(torch_mod, build, nb, r, dim):
    """Bands of two rows and more over fewer than 9 elements.  OpenBLAS's Haswell / Zen build runs the kernels
    it runs for longer rows (model 2 / 1); its SkylakeX build small-matrix kernels of its own - rows in blocks of 16 / 8 / 4 / 2 / 1,
    each with its own arithmetic per length (model 3, found by tools/blas_order/small_matrix_search.py).  CPU tests pin both models
    to NumPy on that build; here every key bit of a hasher pinned to the build is the sign of the model's value, on random rows and
    on rows cancelled against a hyperplane, and - where this host runs that build - the keys are the reference-literal loop's."""
    from lshrs_amd import _hostblas
    from oracle.lshrs_oracle import hash_batch_literal_packed

    torch = torch_mod
    h = _hasher(23, nb, r, dim, reference_blas=build)
    model = h._replay_model()
    assert model == (3 if build == "openblas-skylakex" else (1 if dim % 4 == 0 else 2))
    n = 3_000
    x = np.random.default_rng(dim + r).standard_normal((n, dim)).astype(np.float32)
    stack = np.concatenate([np.asarray(p, dtype=np.float64) for p in h.projections])
    if dim > 1:
        for i in range(0, n, 3):                           # a true tie against one hyperplane
            p = stack[(7 * i) % (nb * r)]
            v = x[i].astype(np.float64)
            x[i] = (v - (v @ p) / (p @ p) * p).astype(np.float32)
    x[5] = 0.0
    flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
    got = h.hash_device(torch.from_numpy(x).cuda(), row_flags=flags).cpu().numpy()
    assert h.last_stats["route"] == ("split+replay" if dim == 8 and model != 3 else "f32+replay"), h.last_stats   # (eight elements: resident-image shapes)
    assert h.last_stats["tie_break_engine"] == "device-replay"
    assert flags[5].item() == 1
    lib = _hostblas.load()
    planes = [np.ascontiguousarray(p, dtype=np.float32) for p in h.projections]
    want = np.zeros((n, nb, h.band_bytes), dtype=np.uint8)
    for i in range(0, n, 2):
        xi = np.ascontiguousarray(x[i])
        for b in range(nb):
            for j in range(r):
                y = lib.lshrs_tb_model_row_dot(planes[b][j].ctypes.data, xi.ctypes.data, dim, model, j, r)
                if y > 0:
                    want[i, b, j >> 3] |= 1 << (j & 7)
    assert np.array_equal(got[::2], want[::2]), int((got[::2] != want[::2]).any(axis=(1, 2)).sum())
    if h._host_blas_agrees():                              # this host's NumPy IS that build: the reference's own bytes
        assert np.array_equal(got, hash_batch_literal_packed(h.projections, x))
    # the default hasher on this host: the device route (both builds are modelled) - the reference-literal loop's bytes
    d = _hasher(23, nb, r, dim)
    gd = d.hash_device(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(gd, hash_batch_literal_packed(d.projections, x)), d.last_stats
    if d._replay_model():
        assert d.last_stats["tie_break_engine"] == "device-replay", d.last_stats
    # a handful of host rows (ingest / query): the same bytes
    small = d.hash_batch_packed(x[:40])
    assert np.array_equal(small, gd[:40])
