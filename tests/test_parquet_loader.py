"""Parquet loader fast path against the literal restatement of the reference loader
(oracle/parquet_oracle.py), and that restatement against the real reference where it is present.  CPU only."""

from __future__ import annotations

import importlib.util
import os
import time

import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")
import pyarrow.parquet as pq  # noqa: E402

from lshrs_amd.parquet_fast import iter_parquet_vectors  # noqa: E402
from oracle.parquet_oracle import iter_parquet_vectors_literal  # noqa: E402


def write(path, ids, vectors_arrow, *, names=("index", "vector"), row_group_size=None):
    table = pa.table({names[0]: ids, names[1]: vectors_arrow})
    pq.write_table(table, path, row_group_size=row_group_size)
    return path


def same(a, b):
    a, b = list(a), list(b)
    assert len(a) == len(b)
    for (ia, va), (ib, vb) in zip(a, b):
        assert ia == ib and all(type(i) is int for i in ia)
        assert va.dtype == np.float32 and va.flags.c_contiguous and va.shape == vb.shape
        assert np.array_equal(va, vb)


@pytest.mark.parametrize("kind", ["list_f32", "list_f64", "fixed_f32", "large_list_f32", "list_int", "list_f16"])
def test_fast_path_equals_reference_restatement(tmp_path, kind):
    rng = np.random.default_rng(5)
    n, dim = 2500, 24
    data = rng.standard_normal((n, dim))
    ids = pa.array(rng.permutation(10**6)[:n], type=pa.int64())
    if kind == "list_f32":
        col = pa.array(data.astype(np.float32).tolist(), type=pa.list_(pa.float32()))
    elif kind == "list_f64":
        col = pa.array(data.tolist(), type=pa.list_(pa.float64()))
    elif kind == "fixed_f32":
        col = pa.FixedSizeListArray.from_arrays(pa.array(data.astype(np.float32).reshape(-1)), dim)
    elif kind == "large_list_f32":
        col = pa.array(data.astype(np.float32).tolist(), type=pa.large_list(pa.float32()))
    elif kind == "list_int":
        col = pa.array(rng.integers(-5, 5, size=(n, dim)).tolist(), type=pa.list_(pa.int32()))
    else:
        col = pa.array(data.astype(np.float16).tolist(), type=pa.list_(pa.float16()))
    path = write(tmp_path / "v.parquet", ids, col, row_group_size=700)     # batches straddle row groups
    for bs in (1000, 10_000, 1):
        if bs == 1 and kind != "list_f32":
            continue
        same(iter_parquet_vectors(path, batch_size=bs), iter_parquet_vectors_literal(path, batch_size=bs))
    renamed = write(tmp_path / "r.parquet", ids, col, names=("id", "emb"))
    same(iter_parquet_vectors(renamed, index_column="id", vector_column="emb"),
         iter_parquet_vectors_literal(renamed, index_column="id", vector_column="emb"))


def test_errors_match_the_reference(tmp_path):
    ids = pa.array([1, 2, 3], type=pa.int64())
    ragged = write(tmp_path / "ragged.parquet", ids, pa.array([[1.0, 2.0], [3.0], [4.0, 5.0]], type=pa.list_(pa.float32())))
    empty = write(tmp_path / "empty.parquet", ids, pa.array([[1.0], [], [2.0]], type=pa.list_(pa.float32())))
    first_empty = write(tmp_path / "fe.parquet", ids, pa.array([[], [1.0], [2.0]], type=pa.list_(pa.float32())))
    nulls = write(tmp_path / "nulls.parquet", ids, pa.array([[1.0, 2.0], None, [3.0, 4.0]], type=pa.list_(pa.float32())))
    good = write(tmp_path / "good.parquet", ids, pa.array([[1.0, 2.0]] * 3, type=pa.list_(pa.float32())))
    for path in (ragged, empty, first_empty, nulls):
        with pytest.raises(Exception) as want:
            list(iter_parquet_vectors_literal(path))
        with pytest.raises(type(want.value)) as got:
            list(iter_parquet_vectors(path))
        assert str(got.value) == str(want.value)
    with pytest.raises(FileNotFoundError, match="does not exist"):
        list(iter_parquet_vectors(tmp_path / "nope.parquet"))
    with pytest.raises(ValueError, match="batch_size"):
        list(iter_parquet_vectors(good, batch_size=0))
    with pytest.raises(ValueError, match="was not found in Parquet schema"):
        list(iter_parquet_vectors(good, vector_column="embedding"))


def test_restatement_equals_the_real_reference_loader(tmp_path):
    ref_file = "/root/reference/lshrs/io/parquet.py"
    if not os.path.exists(ref_file):
        pytest.skip("reference checkout not present (only in the build container)")
    spec = importlib.util.spec_from_file_location("ref_parquet_loader", ref_file)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.default_rng(6)
    ids = pa.array(np.arange(1200), type=pa.int32())
    col = pa.array(rng.standard_normal((1200, 16)).tolist(), type=pa.list_(pa.float64()))
    path = write(tmp_path / "v.parquet", ids, col, row_group_size=500)
    same(iter_parquet_vectors_literal(path, batch_size=400), ref.iter_parquet_vectors(path, batch_size=400))
    same(iter_parquet_vectors(path, batch_size=400), ref.iter_parquet_vectors(path, batch_size=400))


def test_create_signatures_uses_the_fast_loader_and_is_faster(tmp_path, monkeypatch):
    from tests._doubles import make_cpu_lshrs
    from lshrs_amd import InMemoryStorage

    rng = np.random.default_rng(7)
    n, dim = 6000, 128
    data = rng.standard_normal((n, dim)).astype(np.float32)
    path = write(tmp_path / "c.parquet", pa.array(np.arange(n), type=pa.int64()),
                 pa.FixedSizeListArray.from_arrays(pa.array(data.reshape(-1)), dim))
    t0 = time.perf_counter()
    fast = list(iter_parquet_vectors(path, batch_size=2000))
    t1 = time.perf_counter()
    slow = list(iter_parquet_vectors_literal(path, batch_size=2000))
    t2 = time.perf_counter()
    same(fast, slow)
    assert (t1 - t0) < (t2 - t1), "array path should beat the per-value path"
    store = InMemoryStorage()
    idx = make_cpu_lshrs(monkeypatch, dim=dim, num_perm=64, storage=store, packed_ingest=False)
    idx.create_signatures(format="parquet", source=path, batch_size=2000)
    assert store.total_operations == n * 16 and len(store.batches) == 3 * 4   # per loader batch: 3 flushes of 10 000 ops + the final 2 000
