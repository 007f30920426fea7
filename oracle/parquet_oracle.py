"""CPU restatement of the reference's Parquet loader (TEST INFRASTRUCTURE ONLY).

Follows lshrs/io/parquet.py:47-227 (`iter_parquet_vectors`) and :230-320 (`_coerce_vectors`) literally:
``to_pylist()`` for both columns, ``int()`` per id, ``np.asarray(row, float32).reshape(-1)`` per vector,
``np.stack``.  tests/test_parquet_loader.py checks it against the real reference module when that is
present (build container) and uses it to judge lshrs_amd/parquet_fast.py everywhere.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np


def iter_parquet_vectors_literal(source, *, index_column="index", vector_column="vector", batch_size=10_000):
    import pyarrow.parquet as pq

    path = Path(source).expanduser()
    if not path.exists():
        raise FileNotFoundError(f"Parquet source '{path}' does not exist")
    if batch_size <= 0:
        raise ValueError("batch_size must be greater than zero")
    pf = pq.ParquetFile(path)
    schema = pf.schema_arrow
    for column in (index_column, vector_column):
        if schema.get_field_index(column) == -1:
            raise ValueError(f"Column '{column}' was not found in Parquet schema {schema.names}")
    for batch in pf.iter_batches(batch_size=batch_size, columns=[index_column, vector_column]):
        if batch.num_rows == 0:
            continue
        ids = [int(v) for v in batch.column(0).to_pylist()]
        rows, dim = [], None
        for row in batch.column(1).to_pylist():
            arr = np.asarray(row, dtype=np.float32).reshape(-1)
            if arr.size == 0:
                raise ValueError("Encountered empty vector while reading Parquet data")
            if dim is None:
                dim = arr.shape[0]
            elif arr.shape[0] != dim:
                raise ValueError(f"All vectors must share the same dimensionality; expected {dim}, received {arr.shape[0]}")
            rows.append(arr)
        yield ids, np.stack(rows, axis=0)
