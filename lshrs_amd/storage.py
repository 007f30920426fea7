"""Storage seam of the orchestrator.

Redis I/O is OUT OF SCOPE of this build (BASELINE.json north_star: "Redis ... I/O stay
untouched"): a deployment passes the reference's own ``RedisStorage`` instance as
``LSHRS(storage=...)`` and this package only *consumes* its interface
(lshrs/storage/redis.py: ``BucketOperation`` :37, ``bucket_key`` :187, ``add_to_bucket`` :227,
``get_bucket`` :282, ``batch_add`` :348, ``remove_indices`` :419, ``clear`` :590, ``close`` :160).

``InMemoryStorage`` is a dependency-free stand-in with that interface (neither ``redis`` nor
``fakeredis`` exists in the build image); it is what BASELINE config 1 ("fakeredis storage,
plumbing") runs against, and it records every ``batch_add`` batch so tests can assert the
flush boundaries the reference produces.
"""

from __future__ import annotations

import threading
from typing import Dict, Iterable, List, Set, Tuple

BucketOperation = Tuple[int, bytes, int]  # (band_id, band key, vector index) — redis.py:37

__all__ = ["BucketOperation", "InMemoryStorage", "default_storage"]


class InMemoryStorage:
    """Thread-safe dict-of-sets bucket store: ``{prefix}:{band}:bucket:{hex}`` -> set of ids."""

    def __init__(self, *, prefix: str = "lsh", fail_on_flush: bool = False) -> None:
        self.prefix = prefix
        self._buckets: Dict[str, Set[int]] = {}
        self._lock = threading.Lock()
        self._fail_on_flush = fail_on_flush
        self.batches: List[List[BucketOperation]] = []
        self.packed_batches: List[Tuple[int, int]] = []   # (vectors, distinct buckets) per batch_add_packed call
        self.closed = False

    # key format of the reference (redis.py:187-225)
    def bucket_key(self, band_id: int, hash_val: bytes) -> str:
        return f"{self.prefix}:{band_id}:bucket:{bytes(hash_val).hex()}"

    def add_to_bucket(self, band_id: int, hash_val: bytes, index: int) -> None:
        with self._lock:
            self._buckets.setdefault(self.bucket_key(band_id, hash_val), set()).add(int(index))

    def get_bucket(self, band_id: int, hash_val: bytes) -> Set[int]:
        with self._lock:
            return set(self._buckets.get(self.bucket_key(band_id, hash_val), ()))

    def batch_add(self, operations: Iterable[BucketOperation]) -> None:
        ops = list(operations)
        if self._fail_on_flush:
            raise ConnectionError("simulated storage failure")
        with self._lock:
            self.batches.append(ops)
            for band_id, hash_val, index in ops:
                self._buckets.setdefault(self.bucket_key(band_id, hash_val), set()).add(int(index))

    def batch_add_packed(self, ids, keys) -> None:
        """Array form of :meth:`batch_add` (see lshrs_amd/packed_ops.py): ``keys`` is the ``(n, bands, B)``
        uint8 key array, ``ids`` the n vector ids.  Same bucket contents as the equivalent op list."""
        from .packed_ops import group_by_bucket

        if self._fail_on_flush:
            raise ConnectionError("simulated storage failure")
        groups = list(group_by_bucket(ids, keys))
        with self._lock:
            self.packed_batches.append((len(ids), len(groups)))
            for band, key_bytes, members in groups:
                self._buckets.setdefault(self.bucket_key(band, key_bytes), set()).update(members.tolist())

    def remove_indices(self, indices: Iterable[int]) -> None:
        gone = {int(i) for i in indices}
        with self._lock:
            for members in self._buckets.values():
                members -= gone

    def clear(self) -> None:
        with self._lock:
            self._buckets.clear()

    def close(self) -> None:
        self.closed = True

    # conveniences for tests / stats
    @property
    def total_operations(self) -> int:
        with self._lock:
            return sum(len(b) for b in self.batches)

    @property
    def unique_indices(self) -> Set[int]:
        with self._lock:
            return {i for b in self.batches for _, _, i in b}


def default_storage(**redis_kwargs):
    """What ``LSHRS(storage=None)`` does in the reference: build a ``RedisStorage`` from the
    ``redis_*`` arguments (lshrs/core/main.py:232-240).  The Redis client is not part of this
    build, so this resolves the reference's own class when it is importable and fails with a
    clear message otherwise."""
    try:
        from lshrs.storage.redis import RedisStorage  # the untouched reference component
    except Exception as exc:  # pragma: no cover - depends on the deployment
        raise RuntimeError(
            "No storage given and the reference RedisStorage (package `lshrs`, needs `redis`) is not "
            "importable here. Pass storage=<RedisStorage instance> or storage=InMemoryStorage()."
        ) from exc
    return RedisStorage(**redis_kwargs)
