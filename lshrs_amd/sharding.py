"""Multi-GPU ingestion: row-sharding of the signature pass, one process per GPU.

The signature pass is embarrassingly parallel over vectors (the only shared state is the
read-only hyperplane matrix, 0.8-3 MB, replicated on every GPU), so N GPUs of a node each
hash a contiguous row range of every loader batch and write their own bucket operations.
There is NO collective in the data path (BASELINE.json north_star: "replicated hyperplanes,
no collectives"); ``torch.distributed`` is used only for rendezvous/barriers and, on request,
to concatenate the packed keys for verification.  The cosine rerank is single-GPU by design
("replicas only": a second GPU would hold a corpus replica and take disjoint queries).

Launch: ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...`` — backend
``nccl`` (= RCCL over xGMI) on GPUs, ``gloo`` in the CPU tests.
"""

from __future__ import annotations

import os
from typing import Optional, Sequence, Tuple

import numpy as np

__all__ = ["shard_range", "world_info", "index_sharded", "hash_sharded"]


def shard_range(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous balanced split of ``range(n)``: the first ``n % world_size`` ranks get one extra row."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("need 0 <= rank < world_size")
    base, extra = divmod(int(n), world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def world_info() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def index_sharded(lsh, indices: Sequence[int], vectors, *, rank: Optional[int] = None,
                  world_size: Optional[int] = None) -> Tuple[int, int]:
    """Index this rank's slice of a loader batch through ``lsh`` (an ``LSHRS`` bound to this
    rank's GPU).  Every rank passes the same ``(indices, vectors)``; together the ranks index each
    row exactly once.  Returns the ``(lo, hi)`` row range handled here."""
    r, w, _ = world_info()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    lo, hi = shard_range(len(indices), world_size, rank)
    if hi > lo:
        vecs = None if vectors is None else vectors[lo:hi]
        lsh.index(list(indices[lo:hi]), vecs)
    return lo, hi


def hash_sharded(hasher, vectors, *, rank: Optional[int] = None, world_size: Optional[int] = None,
                 gather: bool = False, group=None):
    """Hash this rank's row range of ``vectors`` (host array).  With ``gather=True`` every rank
    also receives the full ``(n, num_bands, band_bytes)`` array, assembled on the host from the
    per-rank pieces (verification / small jobs only — production ingestion writes buckets per rank)."""
    r, w, _ = world_info()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    arr = np.asarray(vectors, dtype=np.float32)
    lo, hi = shard_range(arr.shape[0], world_size, rank)
    local = hasher.hash_batch_packed(arr[lo:hi])
    if not gather or world_size == 1:
        return local
    import torch
    import torch.distributed as dist

    pieces = [None] * world_size
    dist.all_gather_object(pieces, local, group=group)
    return np.concatenate(pieces, axis=0)
