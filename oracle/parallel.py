"""The literal restatement of the reference's hashing loop (``oracle.lshrs_oracle.hash_batch_literal_packed``,
lshrs/hash/lsh.py:162-211) spread over host cores, for full-size parity checks.

TEST INFRASTRUCTURE, like everything under ``oracle/``: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker.  Each worker is a separate (spawned)
process that runs the very per-vector / per-band NumPy calls of the reference on its slice of the rows - the same
library, the same kernel selection, the same operands as one process walking all rows, hence the same bytes.
The vectors travel through one shared-memory block (no pickling of gigabytes).
"""

from __future__ import annotations

import multiprocessing as mp
import os
from multiprocessing import shared_memory
from typing import Optional, Sequence

import numpy as np


def _slice_worker(args):
    name, shape, lo, hi, planes = args
    from oracle.lshrs_oracle import hash_batch_literal_packed

    shm = shared_memory.SharedMemory(name=name)
    try:
        x = np.ndarray(shape, dtype=np.float32, buffer=shm.buf)
        return lo, hash_batch_literal_packed(planes, x[lo:hi])
    finally:
        shm.close()


class SharedVectors:
    """An ``(n, dim)`` float32 matrix in POSIX shared memory (``.array``); unlink with ``close()``."""

    def __init__(self, n: int, dim: int) -> None:
        self.shm = shared_memory.SharedMemory(create=True, size=max(1, n * dim * 4))
        self.shape = (n, dim)
        self.array = np.ndarray(self.shape, dtype=np.float32, buffer=self.shm.buf)

    def close(self) -> None:
        self.array = None
        try:
            self.shm.close()
        finally:
            self.shm.unlink()

    def __enter__(self) -> "SharedVectors":
        return self

    def __exit__(self, *exc) -> None:
        self.close()


def hash_shared_literal_packed(projections: Sequence[np.ndarray], vectors: SharedVectors, workers: Optional[int] = None,
                               slice_rows: int = 8192) -> np.ndarray:
    """Keys of every row of ``vectors`` by the reference-literal loop, ``workers`` processes (default: up to 16)."""
    n, _ = vectors.shape
    planes = [np.ascontiguousarray(p, dtype=np.float32) for p in projections]
    bb = (planes[0].shape[0] + 7) // 8
    out = np.empty((n, len(planes), bb), dtype=np.uint8)
    if n == 0:
        return out
    if workers is None:
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:  # pragma: no cover
            cores = os.cpu_count() or 1
        workers = max(1, min(16, cores))
    jobs = [(vectors.shm.name, vectors.shape, lo, min(n, lo + slice_rows), planes) for lo in range(0, n, slice_rows)]
    ctx = mp.get_context("spawn")       # never fork a process that has initialised the GPU runtime
    with ctx.Pool(processes=workers) as pool:
        for lo, keys in pool.imap_unordered(_slice_worker, jobs):
            out[lo:lo + keys.shape[0]] = keys
    return out
