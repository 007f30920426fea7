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

import numpy as np

BucketOperation = Tuple[int, bytes, int]  # (band_id, band key, vector index) — redis.py:37

__all__ = ["BucketOperation", "InMemoryStorage", "default_storage"]


class InMemoryStorage:
    """Thread-safe dict-of-sets bucket store: bucket ``{prefix}:{band}:bucket:{hex}`` -> set of ids (kept under the pair
    ``(band, key bytes)``; the reference's key text is formatted where somebody asks for it: :meth:`bucket_key`,
    :meth:`bucket_contents`).  ``record_batches=False``: do not keep every ``batch_add`` list alive in ``batches`` (the tests'
    record of the flush boundaries; 100 bytes per operation)."""

    def __init__(self, *, prefix: str = "lsh", fail_on_flush: bool = False, record_batches: bool = True) -> None:
        self.prefix = prefix
        self._buckets: Dict[Tuple[int, bytes], Set[int]] = {}
        self._lock = threading.Lock()
        self._fail_on_flush = fail_on_flush
        self.record_batches = bool(record_batches)
        self.batches: List[List[BucketOperation]] = []
        self.packed_batches: List[Tuple[int, int]] = []   # (vectors, distinct buckets) per batch_add_packed call
        self._segments: list = []                         # BucketCSR of every batch_add_csr call (array-backed buckets)
        self._version = 0                                 # bumped by everything that changes a bucket (array_segments_token)
        self.closed = False

    # key format of the reference (redis.py:187-225)
    def bucket_key(self, band_id: int, hash_val: bytes) -> str:
        return f"{self.prefix}:{band_id}:bucket:{bytes(hash_val).hex()}"

    def add_to_bucket(self, band_id: int, hash_val: bytes, index: int) -> None:
        with self._lock:
            self._version += 1
            self._buckets.setdefault((int(band_id), bytes(hash_val)), set()).add(int(index))

    compact_above = 32     # array segments a lookup tolerates before it folds them into one (many small index() calls)

    def _compact_locked(self) -> None:
        """Fold the array segments into one per key width (lookups cost one bisection per segment)."""
        from .packed_ops import merge_csr

        by_width: Dict[int, list] = {}
        rest = []
        for seg in self._segments:
            (by_width.setdefault(seg.band_bytes, []) if seg.codes is not None else rest).append(seg)
        self._segments = [merge_csr(group) if len(group) > 1 else group[0] for group in by_width.values()] + rest
        self._version += 1

    def get_bucket(self, band_id: int, hash_val: bytes) -> Set[int]:
        with self._lock:
            if len(self._segments) > self.compact_above:
                self._compact_locked()
            out = set(self._buckets.get((int(band_id), bytes(hash_val)), ()))
            segments = list(self._segments)
        key = bytes(hash_val)
        for seg in segments:
            if len(key) != seg.band_bytes or len(seg) == 0:
                continue
            if seg.codes is not None:
                code = (int(band_id) << (8 * seg.band_bytes)) | int.from_bytes(key, "little")
                g = int(np.searchsorted(seg.codes, code))
                hit = g < len(seg) and int(seg.codes[g]) == code
            else:
                cand = np.flatnonzero((seg.bands == band_id) & (seg.key_bytes == np.frombuffer(key, np.uint8)).all(axis=1))
                hit, g = cand.size > 0, int(cand[0]) if cand.size else 0
            if hit:
                out.update(seg.members[seg.offsets[g]:seg.offsets[g + 1]].tolist())
        return out

    @property
    def prefers_batched_lookup(self) -> bool:
        """True once buckets live in array segments (``batch_add_csr``): a query then reads all its bands' buckets
        through :meth:`get_buckets_many` rather than one :meth:`get_bucket` per band."""
        return bool(self._segments)

    def array_segments(self, band_bytes: int):
        """For a device mirror of the index (``lshrs_amd/_query_device.py``): the array segments holding keys of this width -
        a snapshot of the list; the segments themselves are never modified in place, only replaced - provided EVERY bucket of
        the store lives in such segments (no op-tuple buckets, no keys wider than 6 bytes); else None: the caller reads
        buckets through :meth:`get_bucket`.  An empty store gives ``[]``."""
        with self._lock:
            if len(self._segments) > self.compact_above:
                self._compact_locked()
            if any(self._buckets.values()) or band_bytes > 6:
                return None
            segs = [s for s in self._segments if len(s)]
            if any(s.codes is None for s in segs):
                return None
            return [s for s in segs if s.band_bytes == band_bytes]

    def array_segments_token(self):
        """Something that changes whenever :meth:`array_segments` could answer differently - or None where every call must ask
        (buckets kept as dict entries).  A query per call (``LSHRS.get_top_k``) keeps the answer beside the token instead of
        walking the segments again (no lock: two integers and a length)."""
        if self._buckets:
            return None
        return (self._version, id(self._segments), len(self._segments))

    def get_buckets_many(self, keys) -> Tuple[np.ndarray, np.ndarray]:
        """Every member of every bucket a batch of queries touches, as two flat arrays ``(query index, member id)`` - one
        pair per (query, band, member), a member counted ONCE per band however many times and through whichever calls it
        was indexed (buckets are sets) - without a Python object per member (the collision count of
        ``LSHRS._candidate_counts``, lshrs/core/main.py:1101-1109, then is one sort).  ``keys``: (q, bands, B) uint8."""
        from .packed_ops import key_codes

        keys = np.ascontiguousarray(keys, dtype=np.uint8)
        nq, nb, bb = keys.shape
        qs, bs, ms = [], [], []
        with self._lock:
            if len(self._segments) > self.compact_above:
                self._compact_locked()
            segments = [s for s in self._segments if s.band_bytes == bb and len(s)]
            array_path = bb <= 6 and all(seg.codes is not None for seg in segments)
            if self._buckets and array_path:            # buckets built from op tuples: dict lookups per (query, band),
                for qi in range(nq):                    # under the lock (writers mutate these sets)
                    for b in range(nb):
                        mem = self._buckets.get((b, keys[qi, b].tobytes()))
                        if mem:
                            ms.append(np.fromiter(mem, dtype=np.int64, count=len(mem)))
                            qs.append(np.full(len(mem), qi, dtype=np.int64))
                            bs.append(np.full(len(mem), b, dtype=np.int64))
        if not array_path:                              # wide keys: get_bucket merges every source into one set per bucket
            for qi in range(nq):
                for b in range(nb):
                    mem = self.get_bucket(b, keys[qi, b].tobytes())
                    if mem:
                        ms.append(np.fromiter(mem, dtype=np.int64, count=len(mem)))
                        qs.append(np.full(len(mem), qi, dtype=np.int64))
            return (np.concatenate(qs) if qs else np.empty(0, np.int64),
                    np.concatenate(ms) if ms else np.empty(0, np.int64))
        # Set semantics across sources: an id indexed through two calls sits in the same bucket twice - once per source - and
        # must count once.  That needs a sort of every (query, band, member) pair, so it is only paid where it can happen:
        # op-tuple buckets that were hit, a segment that may list a member twice, or segments whose id ranges overlap
        # (sequential ingest - the normal state between compactions - gives disjoint ranges: no id is in two of them).
        tuple_hits, seg_hits, dedupe = bool(qs), 0, False
        spans = sorted(_id_span(seg) for seg in segments)
        overlapping = any(a[1] >= b[0] for a, b in zip(spans[:-1], spans[1:]))
        codes = key_codes(keys).reshape(-1)
        qidx = np.repeat(np.arange(nq, dtype=np.int64), nb)
        bidx = np.tile(np.arange(nb, dtype=np.int64), nq)
        if len(segments) > 0 and codes.shape[0] > 4096:
            # the needles in ascending order: a binary search per needle then walks memory it has just touched (160 000 lookups in
            # a million-code segment: 19 -> 4 ms); every segment reuses the one ordering
            by_code = np.argsort(codes, kind="stable")
            codes, qidx, bidx = codes[by_code], qidx[by_code], bidx[by_code]
        for seg in segments:
            g = np.searchsorted(seg.codes, codes)
            g[g >= len(seg)] = 0
            hit = seg.codes[g] == codes
            g, q, b = g[hit], qidx[hit], bidx[hit]
            lo = seg.offsets[g]
            lens = seg.offsets[g + 1] - lo
            total = int(lens.sum())
            if total == 0:
                continue
            # positions lo[i] .. lo[i] + lens[i] for every hit i, concatenated
            starts = np.cumsum(lens) - lens
            pos = np.arange(total, dtype=np.int64) - np.repeat(starts, lens) + np.repeat(lo, lens)
            ms.append(seg.members[pos])
            qs.append(np.repeat(q, lens))
            bs.append(np.repeat(b, lens))
            seg_hits += 1
            dedupe = dedupe or not seg.distinct         # (a segment that may list a member twice in one bucket)
        if not qs:
            return np.empty(0, np.int64), np.empty(0, np.int64)
        q, m = np.concatenate(qs), np.concatenate(ms)
        if dedupe or (tuple_hits and seg_hits) or (seg_hits > 1 and overlapping):
            # the same id may sit in the same bucket through two sources (indexed twice): once per (query, band)
            b = np.concatenate(bs)
            order = np.lexsort((m, b, q))
            q, b, m = q[order], b[order], m[order]
            keep = np.r_[True, (q[1:] != q[:-1]) | (b[1:] != b[:-1]) | (m[1:] != m[:-1])]
            q, m = q[keep], m[keep]
        return q, m

    def batch_add(self, operations: Iterable[BucketOperation]) -> None:
        ops = list(operations)
        if self._fail_on_flush:
            raise ConnectionError("simulated storage failure")
        with self._lock:
            self._version += 1
            if self.record_batches:
                self.batches.append(ops)
            buckets = self._buckets
            get = buckets.get
            try:        # what LSHRS sends: (int, bytes, int) - the pair in front IS the bucket's name here
                for op in ops:
                    name = op[:2]
                    members = get(name)
                    if members is None:
                        buckets[name] = {op[2]}
                    else:
                        members.add(op[2])
            except TypeError:      # (a bytearray / memoryview key, a NumPy integer: normalised - adding twice is harmless, buckets are sets)
                for band_id, hash_val, index in ops:
                    buckets.setdefault((int(band_id), bytes(hash_val)), set()).add(int(index))

    def batch_add_packed(self, ids, keys) -> None:
        """Array form of :meth:`batch_add` (see lshrs_amd/packed_ops.py): ``keys`` is the ``(n, bands, B)``
        uint8 key array, ``ids`` the n vector ids.  Same bucket contents as the equivalent op list."""
        from .packed_ops import group_by_bucket

        if self._fail_on_flush:
            raise ConnectionError("simulated storage failure")
        groups = list(group_by_bucket(ids, keys))
        with self._lock:
            self._version += 1
            self.packed_batches.append((len(ids), len(groups)))
            for band, key_bytes, members in groups:
                self._buckets.setdefault((int(band), bytes(key_bytes)), set()).update(members.tolist())

    def batch_add_csr(self, csr) -> None:
        """A whole batch's buckets as one :class:`lshrs_amd.packed_ops.BucketCSR`: kept as arrays (an O(1) append; no
        Python object per member or per bucket), consulted by ``get_bucket`` / ``get_buckets_many`` by bisection."""
        if self._fail_on_flush:
            raise ConnectionError("simulated storage failure")
        from .packed_ops import dedupe_csr

        csr = dedupe_csr(csr)                            # (no-op for the builders' output: they mark it distinct)
        with self._lock:
            self._version += 1
            self.packed_batches.append((int(csr.vectors), len(csr)))
            self._segments.append(csr)

    def remove_indices(self, indices: Iterable[int]) -> None:
        gone = {int(i) for i in indices}
        with self._lock:
            self._version += 1
            for members in self._buckets.values():
                members -= gone
            if self._segments:
                from .packed_ops import BucketCSR

                gone_arr = np.fromiter(gone, dtype=np.int64, count=len(gone))
                kept = []
                for seg in self._segments:
                    keep = ~np.isin(seg.members, gone_arr)
                    if keep.all():
                        kept.append(seg)
                        continue
                    lens = np.add.reduceat(keep.astype(np.int64), seg.offsets[:-1]) if len(seg) else np.empty(0, np.int64)
                    lens[seg.offsets[:-1] == seg.offsets[1:]] = 0
                    live = lens > 0
                    kept.append(BucketCSR(seg.band_bytes, seg.bands[live], seg.key_bytes[live],
                                          None if seg.codes is None else seg.codes[live],
                                          np.r_[0, np.cumsum(lens[live])].astype(np.int64), seg.members[keep], seg.vectors,
                                          seg.distinct))
                self._segments = kept

    def clear(self) -> None:
        with self._lock:
            self._version += 1
            self._buckets.clear()
            self._segments = []

    def close(self) -> None:
        self.closed = True

    # conveniences for tests / stats
    def bucket_contents(self) -> Dict[str, Set[int]]:
        """Every non-empty bucket, ``{prefix}:{band}:bucket:{hex}`` -> set of ids (op-tuple buckets and array segments
        merged)."""
        with self._lock:
            out = {self.bucket_key(b, k): set(v) for (b, k), v in self._buckets.items() if v}
            segments = list(self._segments)
        for seg in segments:
            for g in range(len(seg)):
                name = self.bucket_key(int(seg.bands[g]), seg.key_bytes[g].tobytes())
                out.setdefault(name, set()).update(seg.members[seg.offsets[g]:seg.offsets[g + 1]].tolist())
        return out

    @property
    def total_operations(self) -> int:
        with self._lock:
            return sum(len(b) for b in self.batches)

    @property
    def unique_indices(self) -> Set[int]:
        with self._lock:
            return {i for b in self.batches for _, _, i in b}


def _id_span(seg) -> Tuple[int, int]:
    """(smallest, largest) member id of an array segment, remembered on the segment."""
    span = getattr(seg, "_id_span", None)
    if span is None:
        span = (int(seg.members.min()), int(seg.members.max())) if seg.members.size else (0, -1)
        seg._id_span = span
    return span


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
