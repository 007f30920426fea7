"""Parquet loader fast path.

Same contract as the reference's ``iter_parquet_vectors`` (lshrs/io/parquet.py:47-227): a generator of
``(list[int], ndarray (n, dim) float32)`` batches, same keyword arguments, same exceptions.  The reference
materialises every value as a Python object (``to_pylist()`` + one ``np.asarray`` per row, parquet.py:206-227,
230-320) — O(n x dim) objects, which dominates ingestion once hashing takes microseconds.  Here a list column
is taken as Arrow buffers: the flattened child array becomes the ``(n, dim)`` matrix directly (zero-copy for
float32 data), the offsets prove that every row has the same length.  Columns the fast path does not cover
(nulls, non-list vector types) take a row-by-row path that restates the reference's coercion.
"""

from __future__ import annotations

from pathlib import Path
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np

try:  # optional dependency, as in the reference
    import pyarrow as pa
    import pyarrow.parquet as pq
except ImportError:  # pragma: no cover
    pa = None
    pq = None

__all__ = ["iter_parquet_vectors", "DEFAULT_PARQUET_BATCH_SIZE"]

DEFAULT_PARQUET_BATCH_SIZE = 10_000


def _coerce_rows(rows: Sequence[Sequence[float]]) -> np.ndarray:
    """Row-by-row coercion with the reference's checks and messages (parquet.py:230-320)."""
    out: List[np.ndarray] = []
    dim: Optional[int] = None
    for row in rows:
        arr = np.asarray(row, dtype=np.float32).reshape(-1)
        if arr.size == 0:
            raise ValueError("Encountered empty vector while reading Parquet data")
        if dim is None:
            dim = arr.shape[0]
        elif arr.shape[0] != dim:
            raise ValueError(f"All vectors must share the same dimensionality; expected {dim}, received {arr.shape[0]}")
        out.append(arr)
    return np.stack(out, axis=0)


def _list_column_to_matrix(col) -> Optional[np.ndarray]:
    """Arrow list column -> (n, dim) float32 without touching individual values; None if not applicable."""
    if col.null_count:
        return None
    t = col.type
    n = len(col)
    if pa.types.is_fixed_size_list(t):
        dim = t.list_size
        lengths = None
    elif pa.types.is_list(t) or pa.types.is_large_list(t):
        offsets = col.offsets.to_numpy(zero_copy_only=False)
        lengths = np.diff(offsets)
        dim = int(lengths[0]) if n else 0
    else:
        return None
    vt = t.value_type
    if not (pa.types.is_floating(vt) or pa.types.is_integer(vt)):
        return None
    values = col.flatten()          # respects the slice offset of the batch
    if values.null_count:
        return None
    if lengths is not None:
        if (lengths == 0).any():
            first_bad = int(np.flatnonzero(lengths == 0)[0])
            if (lengths[:first_bad] != dim).any():          # the reference scans rows in order: first failure wins
                j = int(np.flatnonzero(lengths[:first_bad] != dim)[0])
                raise ValueError(f"All vectors must share the same dimensionality; expected {dim}, received {int(lengths[j])}")
            raise ValueError("Encountered empty vector while reading Parquet data")
        if (lengths != dim).any():
            j = int(np.flatnonzero(lengths != dim)[0])
            raise ValueError(f"All vectors must share the same dimensionality; expected {dim}, received {int(lengths[j])}")
    elif dim == 0:
        raise ValueError("Encountered empty vector while reading Parquet data")
    flat = values.to_numpy(zero_copy_only=False)
    return np.ascontiguousarray(flat.reshape(n, dim), dtype=np.float32)


def iter_parquet_vectors(source, *, index_column: str = "index", vector_column: str = "vector",
                         batch_size: int = DEFAULT_PARQUET_BATCH_SIZE) -> Iterator[Tuple[List[int], np.ndarray]]:
    if pq is None:
        raise ImportError(
            "pyarrow is required to stream vectors from Parquet files. Install it via `pip install pyarrow`.")
    path = Path(source).expanduser()
    if not path.exists():
        raise FileNotFoundError(f"Parquet source '{path}' does not exist")
    if batch_size <= 0:
        raise ValueError("batch_size must be greater than zero")
    parquet_file = pq.ParquetFile(path)
    schema = parquet_file.schema_arrow
    for column in (index_column, vector_column):
        if schema.get_field_index(column) == -1:
            raise ValueError(f"Column '{column}' was not found in Parquet schema {schema.names}")
    for batch in parquet_file.iter_batches(batch_size=batch_size, columns=[index_column, vector_column]):
        if batch.num_rows == 0:
            continue
        idx_col, vec_col = batch.column(0), batch.column(1)
        if idx_col.null_count == 0 and pa.types.is_integer(idx_col.type):
            indices = idx_col.to_numpy(zero_copy_only=False).tolist()
        else:
            indices = [int(v) for v in idx_col.to_pylist()]
        vectors = _list_column_to_matrix(vec_col)
        if vectors is None:
            vectors = _coerce_rows(vec_col.to_pylist())
        yield indices, vectors
