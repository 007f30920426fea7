"""``LSHRS`` — the caller side of the hot path, batched for the GPU.

Thin counterpart of the reference orchestrator (lshrs/core/main.py) for exactly the
methods that reach the accelerated path: constructor, ``ingest`` / ``index`` /
``create_signatures`` / ``flush`` (signature pass) and ``query`` / ``get_top_k`` /
``get_above_p`` (signature pass + cosine rerank), plus the storage pass-throughs and the on-disk /
pickle persistence of configuration + hyperplanes (same format as the reference).
Same keyword arguments, return types, error types and messages; storage (Redis) and the
loaders (PostgreSQL / Parquet) are the reference's own components and are not re-implemented.

What is different on purpose: ``index()`` hashes a whole loader batch in ONE kernel launch
instead of looping ``ingest()`` per vector (lshrs/core/main.py:514-515), while reproducing
what that loop lets a caller observe —
  * operations are enqueued vector-major, band-minor (main.py:1125-1128);
  * the buffer is flushed, whole, at the first vector boundary where it holds at least
    ``buffer_size`` operations (main.py:1131-1143), and once more at the end (main.py:518);
  * a bad row (negative id, zero vector) raises the same ``ValueError`` after every earlier row
    was enqueued (and possibly flushed), and the trailing flush is then not reached.
tests/golden/g5_orchestration.json holds the reference's own batches for these cases.
"""

from __future__ import annotations

import itertools
import json
import logging
import math
import warnings
from pathlib import Path
from threading import Lock
from typing import Any, Callable, Dict, Iterable, Iterator, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _hostblas
from ._gcpause import gc_paused as _gc_paused
from .bandrows import get_optimal_config
from .hasher import LSHHasher
from .packed_ops import bucket_csr as _bucket_csr
from .similarity import rerank_padded_arrays as _rerank_padded
from .similarity import top_k_cosine
from .storage import BucketOperation, default_storage

logger = logging.getLogger(__name__)

VectorFetchFn = Callable[[Sequence[int]], np.ndarray]
Loader = Callable[..., Iterator[Tuple[Sequence[int], np.ndarray]]]

__all__ = ["LSHRS", "lshrs"]

_FORMAT_VERSION = "0.1.1a4"  # on-disk format version string the reference writes (main.py:882)
_ZERO_MSG = "Cannot index zero vector - norm undefined. Check embeddings for corruption."


def _device_tensor(obj):
    """`obj` if it is a torch tensor on a GPU (torch is only looked at when it is already imported), else None."""
    import sys

    torch = sys.modules.get("torch")
    if torch is not None and isinstance(obj, torch.Tensor) and obj.is_cuda:
        return obj
    return None


class ReferenceBlasMismatch(UserWarning):
    """An index hashed with ``reference_blas="host"`` is being loaded on a host whose BLAS sums differently at its shape."""


def _check_recorded_blas(reference_blas: str, recorded: Optional[str], rows_per_band: int, dim: int, strict: bool) -> None:
    """The reference's keys are ``sign(P_band @ v)`` AS THE HOST'S ``sgemv`` ROUNDS IT (lshrs/hash/lsh.py:200): an index
    built with ``reference_blas="host"`` on one OpenBLAS build and queried from the other hashes tied projections differently
    wherever the two builds sum differently - a scalar tail (``dim % 4 != 0``), one-row bands, fewer than 9 elements
    (``_hostblas.builds_differ``).  ``recorded`` is the build ``save_to_disk`` / pickle wrote down.  Says so (a
    :class:`ReferenceBlasMismatch` warning naming both builds; ``strict``: ``ValueError``); silent where the builds agree."""
    if reference_blas != "host" or not recorded or not _hostblas.builds_differ(rows_per_band, dim):
        return
    here = _hostblas.host_build_name()
    if here == recorded:
        return
    msg = (f"this index was hashed with reference_blas='host' on a {recorded!r} host; this host's NumPy runs "
           f"{here!r}" + ("" if here else " (a BLAS whose summation order is not recognised)") +
           f", which sums bands of {rows_per_band} rows over {dim} elements differently: keys of near-tie projections may "
           f"differ from the stored ones.  Load with reference_blas={recorded!r} to hash as the index was built.")
    if strict:
        raise ValueError(msg)
    warnings.warn(msg, ReferenceBlasMismatch, stacklevel=3)


class _DeferredStorage:
    """What an unpickled index holds until somebody touches the storage: the reference builds a ``RedisStorage`` inside
    ``__setstate__`` (lazy TCP, lshrs/core/main.py:1010-1044); here the client may not even be importable in the
    process that unpickles (a GPU worker that only hashes), so the storage is resolved on first use."""

    def __init__(self, redis_config: Dict[str, Any]) -> None:
        object.__setattr__(self, "_cfg", dict(redis_config))
        object.__setattr__(self, "_real", None)

    def _resolve(self):
        if self._real is None:
            rc = self._cfg
            object.__setattr__(self, "_real", default_storage(
                host=rc["host"], port=rc["port"], db=rc["db"], password=rc["password"],
                decode_responses=rc["decode_responses"], prefix=rc["prefix"], max_connections=rc["max_connections"]))
        return self._real

    def __getattr__(self, item):
        if item in ("_real", "_cfg") or (item.startswith("__") and item.endswith("__")):
            raise AttributeError(item)          # (copy / pickle probe an instance whose __init__ has not run: no recursion)
        if item in ("batch_add_csr", "batch_add_packed", "get_buckets_many") and self._real is None:
            raise AttributeError(item)          # (capability probes must not open a connection)
        return getattr(self._resolve(), item)


def _ragged_positions(starts: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """Concatenation of ``arange(starts[i], starts[i] + lens[i])`` over i."""
    total = int(lens.sum())
    return np.arange(total, dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens) + np.repeat(starts, lens)


def _split_rows(flat: list, lens: np.ndarray) -> List[list]:
    """A flat Python list cut into consecutive pieces of the given lengths."""
    ends = np.cumsum(lens).tolist()
    out, lo = [], 0
    for hi in ends:
        out.append(flat[lo:hi])
        lo = hi
    return out


class LSHRS:
    """Redis-backed LSH index whose hashing and reranking run on MI355X.

    Keyword arguments are those of the reference constructor (lshrs/core/main.py:154-173).
    Extras: ``hasher`` (inject a ready hasher object; default builds ``LSHHasher``), ``device`` (GPU index for
    the default hasher) and ``packed_ingest``: how ``index()`` hands a batch's buckets to the storage.
    ``"auto"`` (default): batches of at least ``packed_auto_min_ops`` operations go as ONE bucket CSR grouped on the
    device (``storage.batch_add_csr`` / ``batch_add_packed``; the reference's ``RedisStorage`` - anything with its
    ``pipeline()`` + ``bucket_key()`` - is wrapped in ``RedisPackedWriter``: one ``SADD`` per bucket) where the storage
    can take that, smaller batches and other storages get the reference's ``(band, key, id)`` tuples through
    ``batch_add``; ``True``: the array path whenever the storage can take it; ``False``: always the reference's
    operation lists, flush boundaries included (what tests/golden/g5_orchestration.json pins).  Same bucket contents
    every way (lshrs/core/main.py:1113-1143, lshrs/storage/redis.py:348-416).
    ``reference_blas``: which BLAS build the band keys are the reference's keys on (``LSHHasher``; "host" = this process's
    NumPy).  A named build is stored by ``save_to_disk`` (key ``lshrs_amd`` of metadata.json, which the reference's loader
    does not read) and by pickle, so an index and everything that queries it hash alike on any machine.
    """

    def __init__(
        self,
        *,
        dim: int,
        num_perm: int = 128,
        num_bands: Optional[int] = None,
        rows_per_band: Optional[int] = None,
        similarity_threshold: float = 0.5,
        buffer_size: int = 10_000,
        vector_fetch_fn: Optional[VectorFetchFn] = None,
        storage: Any = None,
        redis_host: str = "localhost",
        redis_port: int = 6379,
        redis_db: int = 0,
        redis_password: Optional[str] = None,
        redis_prefix: str = "lsh",
        redis_max_connections: int = 50,
        decode_responses: bool = False,
        seed: int = 42,
        hasher: Any = None,
        device: Any = None,
        packed_ingest: Union[bool, str] = "auto",
        devices: Optional[Sequence[int]] = None,
        reference_blas: str = "host",
    ) -> None:
        if dim <= 0:
            raise ValueError("Vector dimensionality must be greater than zero")
        if num_perm <= 0:
            raise ValueError("num_perm must be greater than zero")
        if buffer_size <= 0:
            raise ValueError("buffer_size must be greater than zero")

        if num_bands is None or rows_per_band is None:
            num_bands, rows_per_band = get_optimal_config(num_perm, similarity_threshold)
        if num_bands is None or rows_per_band is None:  # pragma: no cover - defensive, as the reference
            raise RuntimeError(
                f"Auto-config failed: get_optimal_config({num_perm}, {similarity_threshold}) "
                f"-> ({num_bands}, {rows_per_band})")
        if num_bands * rows_per_band != num_perm:
            raise ValueError(
                f"num_bands * rows_per_band must equal num_perm (received {num_bands} * {rows_per_band} != {num_perm})")

        self._dim = dim
        self._buffer_size = buffer_size
        self._vector_fetch_fn = vector_fetch_fn
        if packed_ingest not in (True, False, "auto"):
            raise ValueError("packed_ingest must be True, False or 'auto'")
        self._packed_ingest = packed_ingest
        self.packed_auto_min_ops = 8_192     # "auto": below this many (band, key, id) operations the tuples are cheaper
        self._packed_writer = None
        self._hasher = hasher if hasher is not None else LSHHasher(
            num_bands=num_bands, rows_per_band=rows_per_band, dim=dim, seed=seed, device=device, devices=devices,
            reference_blas=reference_blas)
        self._storage = storage if storage is not None else default_storage(
            host=redis_host, port=redis_port, db=redis_db, password=redis_password,
            decode_responses=decode_responses, prefix=redis_prefix, max_connections=redis_max_connections)
        self._buffer: List[BucketOperation] = []
        self._buffer_lock = Lock()
        self._corpus = None                  # device-resident vectors for the rerank (set_corpus)
        self.last_query_stats: Dict[str, Any] = {}
        from ._query_device import DeviceBuckets

        self._dev_buckets = DeviceBuckets()  # device mirror of the store's bucket arrays (query_many)
        self._one_query: Dict[int, Any] = {}    # per device: the pinned / device buffers of the single-query chain
        self._one_table = None          # (store token, band bytes, device) -> the segments' descriptor on the device (_query_one_device)
        self._config: Dict[str, Any] = {
            "dim": dim, "num_perm": num_perm, "num_bands": num_bands, "rows_per_band": rows_per_band,
            "similarity_threshold": similarity_threshold, "buffer_size": buffer_size, "seed": seed,
        }
        self._redis_config: Dict[str, Any] = {
            "host": redis_host, "port": redis_port, "db": redis_db, "password": redis_password,
            "prefix": redis_prefix, "decode_responses": decode_responses, "max_connections": redis_max_connections,
        }

    # ------------------------------------------------------------------ lifecycle
    def close(self) -> None:
        self.flush()
        self._storage.close()

    def __enter__(self) -> "LSHRS":
        return self

    def __exit__(self, exc_type, exc_value, traceback) -> None:
        self.close()

    # ------------------------------------------------------------------ ingestion
    def create_signatures(self, format: str = "postgres", **loader_kwargs: Any) -> None:
        """Stream ``(indices, vectors)`` batches from a loader and index each with one launch
        (reference: main.py:315-384).  ``format`` is "postgres"/"pg", "parquet"/"pq" (the reference's
        loaders, when that package is importable) or "batches" with ``batches=<iterable>``."""
        loader = self._resolve_loader(format)
        ingest = None
        try:
            for indices, vectors in loader(**loader_kwargs):
                if ingest is None and len(indices) and self._streams_buckets(len(indices) * self._config["num_bands"]):
                    # whole loader batches, dealt round-robin to one lane per device, their buckets handed to the storage in
                    # batch order (lshrs_amd/_ingest.py): the next batch is copied and hashed while this one's buckets are
                    # grouped and stored
                    self.flush()
                    ingest = self._open_ingest()
                    ingest.__enter__()
                if ingest is None:
                    self.index(indices, vectors)
                    continue
                if len(indices) == 0:
                    continue
                if vectors is None:
                    vectors = self._require_vector_fetch_fn()(indices)
                id_arr, arr = self._check_batch(indices, vectors)
                ingest.submit(id_arr, arr)
                if ingest.failed:
                    break
        except BaseException as exc:
            if ingest is not None:
                # what was handed over is stored as the reference's sequential loop would have stored it - unless the caller is
                # being interrupted: then nothing is waited for beyond the units in flight.  A bad row in a unit already handed
                # over comes first in row order and is what is raised - with the loader's own exception chained behind it.
                try:
                    ingest.close(wait=not isinstance(exc, (KeyboardInterrupt, SystemExit)))
                except BaseException as earlier:  # noqa: BLE001
                    if earlier is not exc:
                        raise earlier from exc
            raise
        if ingest is not None:
            ingest.close()

    def ingest(self, index: int, vector) -> None:
        """Hash one vector and buffer its bucket operations (reference: main.py:386-411)."""
        if index < 0:
            raise ValueError("index must be non-negative")
        arr = self._check_dim(vector)
        keys, flag = self._hash_one(arr)
        if flag & 1:
            raise ValueError(_ZERO_MSG)
        self._enqueue_packed(int(index), keys)
        self._flush_buffer_if_needed()

    def index(self, indices: Sequence[int], vectors: Optional[np.ndarray] = None) -> None:
        """Index a batch: one signature-pass launch, then the reference's enqueue/flush sequence
        (reference: main.py:442-518)."""
        if len(indices) == 0:
            return
        if vectors is None:
            vectors = self._require_vector_fetch_fn()(indices)
        resident = _device_tensor(vectors)
        if resident is not None:
            # vectors that already live on a GPU (round 6): hashed where they are, their buckets grouped on the device - only the
            # bucket arrays cross the link (a store that takes arrays; any other store gets the host form of the same rows)
            id_arr, resident = self._check_batch(indices, resident)
            if self._streams_buckets(int(resident.shape[0]) * self._config["num_bands"]):
                self.flush()
                with self._open_ingest(inline=True) as ingest:
                    ingest.submit(id_arr, resident)
                return
            vectors = resident.detach().cpu().numpy()
        arr = np.asarray(vectors, dtype=np.float32)
        if arr.ndim != 2 or arr.shape[1] != self._dim:
            raise ValueError(f"Vectors must have shape (n, {self._dim}); received {arr.shape}")
        if arr.shape[0] != len(indices):
            raise ValueError(
                "Number of vectors does not match number of indices "
                f"(received {arr.shape[0]} vectors for {len(indices)} indices)")

        if self._streams_buckets(arr.shape[0] * self._config["num_bands"]):
            # array path, pipelined (lshrs_amd/_ingest.py): copy + signature pass chunk by chunk, every chunk's
            # keys grouped into buckets on the device under the next chunk's copy, bucket arrays to the storage in row order.
            # Anything already buffered goes first so the storage sees operations in the original order.
            self.flush()
            id_arr, arr = self._check_batch(indices, arr)
            lanes = self._ingest_hashers()
            with self._open_ingest(inline=True) as ingest:
                if len(lanes) == 1 or arr.shape[0] < 2 * self.lane_rows:
                    ingest.submit(id_arr, arr)
                else:                       # several devices: contiguous slices of whole chunks, dealt round-robin
                    for lo in range(0, arr.shape[0], self.lane_rows):
                        ingest.submit(id_arr[lo:lo + self.lane_rows], arr[lo:lo + self.lane_rows])
            return
        sink = self._packed_sink(arr.shape[0] * self._config["num_bands"])
        packed = sink is not None
        if packed:
            id_arr = np.asarray(indices)
            id_arr = id_arr.astype(np.int64) if id_arr.dtype.kind in "iuf" else np.array([int(i) for i in indices], dtype=np.int64)
            ids = id_arr
        else:
            ids = [int(i) for i in indices]
            id_arr = None
        keys, flags = self._hasher.hash_batch_packed(arr, return_row_flags=True)

        # first row the per-vector loop of the reference would have choked on, and why
        stop, error = len(ids), None
        if id_arr is not None:
            negs = np.flatnonzero(id_arr < 0)
            neg = int(negs[0]) if negs.size else None
        else:
            neg = next((j for j, i in enumerate(ids) if i < 0), None)
        zero_rows = np.flatnonzero(flags & 1)
        zero = int(zero_rows[0]) if zero_rows.size else None
        if neg is not None and (zero is None or neg <= zero):
            stop, error = neg, ValueError("index must be non-negative")
        elif zero is not None:
            stop, error = zero, ValueError(_ZERO_MSG)

        if packed:
            # array path: same buckets, same members, no per-operation Python objects.
            # Anything already buffered goes first so the storage sees operations in the original order.
            self.flush()
            if hasattr(sink, "batch_add_csr"):
                # the whole batch (the rows in front of a bad one) as ONE bucket CSR, grouped on the device
                if stop:
                    sink.batch_add_csr(_bucket_csr(id_arr[:stop], keys[:stop]))
            else:
                per_call = max(1, -(-self._buffer_size // keys.shape[1]))  # vectors per storage call ~ buffer_size ops
                for lo in range(0, stop, per_call):
                    hi = min(stop, lo + per_call)
                    sink.batch_add_packed(id_arr[lo:hi], keys[lo:hi])
            if error is not None:
                raise error
            return

        # The reference's operation tuples (main.py:1113-1143), a flush WINDOW at a time: the buffer is flushed, whole, at the
        # first vector boundary where it holds at least `buffer_size` operations - so the vectors up to that boundary are known
        # from the buffer's length at the window's start, and their operations are built in one pass each over the key bytes
        # (every band key as a `bytes` object), the ids (each `num_bands` times) and the band numbers - zipped, not looped -
        # and appended under ONE lock acquisition.  Same tuples, same order, same flush boundaries as the per-vector loop
        # (tests/golden/g5_orchestration.json); 58 k -> 400 k+ vectors/s of host work at 16 bands.
        nb, bb = int(keys.shape[1]), int(keys.shape[2])
        with _gc_paused():
            key_objs = np.ascontiguousarray(keys[:stop]).reshape(-1, bb).view(np.dtype((np.void, bb)))[:, 0].tolist()
            id_objs = list(itertools.chain.from_iterable(zip(*([ids[:stop]] * nb))))      # every id num_bands times, the SAME int objects
        band_objs = list(range(nb))
        j = 0
        while j < stop:
            with self._buffer_lock:
                held = len(self._buffer)
            take = min(stop - j, max(1, -(-(self._buffer_size - held) // nb)))
            lo, hi = j * nb, (j + take) * nb
            with _gc_paused():
                ops = list(zip(band_objs * take, key_objs[lo:hi], id_objs[lo:hi]))
            with self._buffer_lock:
                self._buffer.extend(ops)
                full = len(self._buffer) >= self._buffer_size
            if full:
                self.flush()
            j += take
        if error is not None:
            raise error
        self.flush()

    def flush(self) -> None:
        """Send every buffered operation in one ``batch_add``; on failure put them back in front
        and re-raise (reference: main.py:413-440)."""
        with self._buffer_lock:
            if not self._buffer:
                return
            pending = self._buffer
            self._buffer = []
        try:
            self._storage.batch_add(pending)
        except Exception as exc:
            logger.error(f"Failed to flush buffer to Redis: {exc}")
            with self._buffer_lock:
                self._buffer[0:0] = pending
            raise

    # ------------------------------------------------------------------ queries
    def query(self, vector, *, top_k: Optional[int] = 10, top_p: Optional[float] = None
              ) -> Union[List[int], List[Tuple[int, float]]]:
        """Band-collision candidates, optionally reranked by cosine (reference: main.py:524-658)."""
        query_vector = self._check_dim(vector)
        answered = self._query_one_device(query_vector, top_k, top_p)
        if answered is not None:
            return answered
        keys, flag = self._hash_one(query_vector)
        if flag & 1:
            raise ValueError(_ZERO_MSG)
        if getattr(self._storage, "prefers_batched_lookup", False):
            # array-backed buckets: one vectorised lookup for all bands and two sorts instead of a Python loop per member
            candidate_indices = self._ordered_candidates_arrays(keys[None])[0].tolist()
        else:
            counts = self._candidate_counts_from_keys(keys)
            candidate_indices = [idx for idx, _ in sorted(counts.items(), key=lambda item: (-item[1], item[0]))]
        if not candidate_indices:
            return []

        if top_p is None:
            if top_k is None:
                top_k = len(candidate_indices)
            if top_k <= 0:
                raise ValueError("top_k must be greater than zero when provided")
            return candidate_indices[:top_k]

        if not 0 < top_p <= 1:
            raise ValueError("top_p must be within the range (0, 1]")
        if self._corpus is not None:
            # the indexed vectors are resident on the device (set_corpus): gathered and scored there, nothing fetched
            from .similarity import rerank_batch

            ranked = rerank_batch(query_vector[None], self._corpus, np.asarray([candidate_indices], dtype=np.int64),
                                  k=len(candidate_indices))[0]
        else:
            arr = self._fetch_checked(self._require_vector_fetch_fn(), candidate_indices)
            ranked = top_k_cosine(query_vector, arr, k=len(candidate_indices))
        scored = [(candidate_indices[pos], score) for pos, score in ranked]
        limit = max(1, math.ceil(len(scored) * top_p))
        if top_k is not None:
            if top_k <= 0:
                raise ValueError("top_k must be greater than zero when provided")
            limit = min(limit, top_k)
        return scored[:limit]

    def _query_one_device(self, query_vector: np.ndarray, top_k, top_p):
        """:meth:`query` as ONE chain of launches with one wait at its end (``_query_device.OneQuery``: signature kernel, bucket
        lookup, collision count and order, [rerank on the attached corpus], cut - the answer straight into pinned memory), where
        the store keeps its buckets as arrays and the hasher's one-launch kernel serves the shape; None: the caller takes the
        host-counted path (any other store or hasher, a rerank through ``vector_fetch_fn``, a list beyond the kernels' capacity).
        Errors in the reference's order: zero vector, then - only if there are candidates - the arguments."""
        from . import _query_device as qd

        h = self._hasher
        try:
            if not qd.OneQuery.applies(h):
                return None
        except Exception:      # noqa: BLE001 - a hasher of another kind
            return None
        bad_p = top_p is not None and not 0 < top_p <= 1
        rerank = top_p is not None and not bad_p
        corpus = self._corpus
        if rerank:
            torch = __import__("torch")
            if not (isinstance(corpus, torch.Tensor) and corpus.is_cuda and corpus.dtype == torch.float32 and corpus.dim() == 2
                    and int(corpus.shape[1]) == self._dim and corpus.stride(1) == 1):
                return None
        st = self._storage
        if isinstance(st, _DeferredStorage):
            st = st._resolve()
        dev = corpus.device if rerank else h._torch_device()
        # what the store's segments look like on the device: kept beside the store's change token (one comparison per call
        # instead of a walk over the segments under the store's lock)
        token_fn = getattr(st, "array_segments_token", None)
        token = token_fn() if callable(token_fn) else None
        kept = self._one_table
        if token is not None and kept is not None and kept[0] == (id(st), token, h.band_bytes, dev.index):
            desc, nseg, max_id = kept[1]
        else:
            segs = st.array_segments(h.band_bytes) if callable(getattr(st, "array_segments", None)) else None
            if segs is None:
                return None
            try:
                desc, nseg, max_id = self._dev_buckets.table(segs, dev)
            except qd.TooLarge:
                return None
            token = token_fn() if callable(token_fn) else None        # (array_segments may have folded the segments)
            # (with it: the segments and the mirror's device arrays the descriptor points into - alive as long as the entry is)
            self._one_table = None if token is None else ((id(st), token, h.band_bytes, dev.index), (desc, nseg, max_id), segs,
                                                          self._dev_buckets._table, st)      # (and the store: its id is the key)
        one = self._one_query.get(dev.index)
        if one is None or one.shape != (h.num_bands, h.band_bytes, h.dim):
            one = self._one_query[dev.index] = qd.OneQuery(h, dev)
        try:
            k_arg = top_k if (top_k is not None and top_k > 0) else -1
            ucount, ids, scores, flag = one.run(h, query_vector, desc, nseg, max_id, k_arg, float(top_p) if rerank else -1.0, corpus)
        except qd.TooLarge:
            return None
        if flag & 1:
            raise ValueError(_ZERO_MSG)
        if ucount < 0:                       # more pairs than the chain's fixed capacity: the batch form, which sizes its arrays
            try:                             # by what it finds (and takes lists beyond the LDS network through global memory)
                got = self._query_many_device(query_vector[None], top_k if (top_k is None or top_k > 0) else None,
                                              top_p if rerank else None, corpus if rerank else None)
            except qd.TooLarge:
                return None
            ids, scores = got[0], got[1]
            ucount = 1 if len(ids) else 0
        if ucount == 0:
            return []
        if bad_p:
            raise ValueError("top_p must be within the range (0, 1]")
        if top_k is not None and top_k <= 0:
            raise ValueError("top_k must be greater than zero when provided")
        if scores is None:
            return ids.tolist()
        return list(zip(ids.tolist(), scores.astype(np.float64).tolist()))

    def get_top_k(self, vector, topk: int = 10) -> List[int]:
        return list(self.query(vector, top_k=topk, top_p=None))  # type: ignore[arg-type]

    def get_above_p(self, vector, p: float = 0.95) -> List[Tuple[int, float]]:
        return list(self.query(vector, top_k=None, top_p=p))  # type: ignore[arg-type]

    def set_corpus(self, corpus) -> None:
        """Attach the indexed vectors as a device-resident ``(m, dim)`` float32 tensor whose row ``i`` is the vector of id
        ``i``: ``query`` / ``get_above_p`` / ``query_many`` then gather their candidates from it on the device instead of
        calling ``vector_fetch_fn`` (lshrs/core/main.py:629-646 fetches and stacks them on the host).  ``None`` detaches."""
        if corpus is not None and (getattr(corpus, "ndim", 0) != 2 or int(corpus.shape[1]) != self._dim):
            raise ValueError(f"corpus must have shape (m, {self._dim})")
        self._corpus = corpus

    def query_many(self, vectors, *, top_k: Optional[int] = 10, top_p: Optional[float] = None, corpus=None,
                   return_arrays: bool = False, engine: str = "auto"):
        """Batched :meth:`query`: returns ``[query(v, top_k=top_k, top_p=top_p) for v in vectors]`` with ONE
        signature launch for all queries, the collision counting and candidate ordering of all of them on the device
        (``lshrs_amd/_query_device.py``: bucket lookup in the device-resident bucket arrays, sort / count / order per query
        inside a workgroup's LDS), ONE rerank over all candidate lists and ONE copy back (reference per
        query: main.py:524-658, ``_candidate_counts`` :1088-1111).

        ``vectors``: ``(n, dim)`` array-like as in the reference - or a torch tensor that already lives on a GPU (round 6: the
        queries then never cross the link; 10 000 x 768 are 30 MB = 0.7 of the 2.7 ms a reranked batch takes).
        ``corpus``: optional device-resident ``(m, dim)`` float32 tensor whose row ``i`` is the vector of id
        ``i`` (default: what :meth:`set_corpus` attached); with it the candidates are gathered on the device and
        ``vector_fetch_fn`` is not called.
        ``return_arrays``: ``(ids, scores, bounds)`` instead of lists - query ``i``'s answer is ``ids[bounds[i]:bounds[i + 1]]``
        (int64) with ``scores[...]`` (float32; ``None`` without ``top_p``): no Python object per result.
        ``engine``: "auto" (the device path wherever the hasher is the HIP one; a batch with a candidate list beyond the
        kernels' 16 384 entries is counted on the host), "device" (raise instead), "host" (NumPy counting between the two
        launches: round 5's path)."""
        resident = _device_tensor(vectors)         # queries that already live on a GPU stay there (no copy over the link, no host array)
        if resident is not None:
            if resident.dim() != 2 or int(resident.shape[1]) != self._dim:
                raise ValueError(f"Vectors must have shape (n, {self._dim}); received {tuple(resident.shape)}")
            if resident.dtype != __import__("torch").float32:
                resident = resident.float()
            arr = None
            shape0 = int(resident.shape[0])
        else:
            arr = np.asarray(vectors, dtype=np.float32)
            if arr.ndim != 2 or arr.shape[1] != self._dim:
                raise ValueError(f"Vectors must have shape (n, {self._dim}); received {arr.shape}")
            shape0 = arr.shape[0]
        if top_p is None and top_k is not None and top_k <= 0:
            raise ValueError("top_k must be greater than zero when provided")
        if top_p is not None and not 0 < top_p <= 1:
            raise ValueError("top_p must be within the range (0, 1]")
        if top_p is not None and top_k is not None and top_k <= 0:
            raise ValueError("top_k must be greater than zero when provided")
        if engine not in ("auto", "device", "host"):
            raise ValueError("engine must be 'auto', 'device' or 'host'")
        if corpus is None:
            corpus = self._corpus
        nq = shape0
        if nq == 0:
            empty = (np.empty(0, np.int64), None if top_p is None else np.empty(0, np.float32), np.zeros(1, np.int64))
            return empty if return_arrays else []
        on_device = (engine != "host" and callable(getattr(self._hasher, "hash_device", None))
                     and getattr(self._hasher, "tie_break", "host") == "host")
        if engine == "device" and not on_device:
            raise RuntimeError("engine='device' needs the HIP hasher")
        got = None
        if on_device:
            from ._query_device import TooLarge

            try:
                got = self._query_many_device(arr if resident is None else resident, top_k, top_p, corpus)
            except TooLarge:
                if engine == "device":
                    raise
        if got is None:
            got = self._query_many_host(arr if resident is None else resident.cpu().numpy(), top_k, top_p, corpus)
        ids, scores, bounds = got
        if return_arrays:
            return ids, scores, bounds
        keep = np.diff(bounds)
        with _gc_paused():
            if scores is None:
                return _split_rows(ids.tolist(), keep)
            return _split_rows(list(zip(ids.tolist(), scores.astype(np.float64).tolist())), keep)

    def _query_many_device(self, arr: np.ndarray, top_k, top_p, corpus):
        """``query_many`` with everything between the upload of the queries and the download of the answers on the device."""
        from . import _native
        from . import _query_device as qd

        torch = _native.require_gpu()
        nq = int(arr.shape[0])
        resident = isinstance(arr, torch.Tensor)
        if not resident and arr.strides[0] < arr.shape[1] * 4:            # (`v[None]`: NumPy gives the new axis stride 0 - the kernels take a row stride)
            arr = arr.reshape(-1).copy().reshape(nq, arr.shape[1])
        if corpus is not None and isinstance(corpus, torch.Tensor) and corpus.is_cuda:
            dev = corpus.device
        elif resident:
            dev = arr.device
        else:
            dev = self._hasher._torch_device()
        with torch.cuda.device(dev):
            if resident:                                # queries handed over on a GPU: rows of `dim` floats, on the device that ranks them
                x = arr.to(dev) if arr.device != dev else arr
                if x.stride(1) != 1 or x.stride(0) < x.shape[1]:
                    x = x.contiguous()
            else:
                x = qd.upload(torch, arr, dev)
            flags = torch.empty(nq, dtype=torch.uint8, device=dev)
            keys_dev = self._hasher.hash_device(x, row_flags=flags)
            if bool((flags & 1).any()):
                raise ValueError(_ZERO_MSG)
            lists = self._device_lists(qd, keys_dev, dev)
            # (diagnostics of the last device-counted batch: bench.py's candidates/s; not part of the answer)
            self.last_query_stats = {"queries": nq, "pairs": lists.total, "longest_list": lists.max_pairs, "engine": "device",
                                     "segments": lists.segments}
            if top_p is None:
                return qd.rank_and_cut(lists, top_k, None)
            if lists.total == 0:
                return np.empty(0, np.int64), np.empty(0, np.float32), np.zeros(nq + 1, np.int64)
            if corpus is not None:
                table = corpus if isinstance(corpus, torch.Tensor) else qd.upload(torch, np.asarray(corpus, dtype=np.float32), dev)
                if table.dtype != torch.float32 or table.dim() != 2 or int(table.shape[1]) != self._dim:
                    raise ValueError(f"corpus must be a float32 tensor of shape (m, {self._dim})")
                if not table.is_cuda:
                    table = table.to(dev)
                if table.stride(1) != 1:
                    table = table.contiguous()
                return qd.rank_and_cut(lists, top_k, top_p, queries_dev=x, corpus=table)
            # no resident corpus: the candidates' vectors come from the caller's fetch function, list by list as the reference
            # asks for them (main.py:629), and travel to the device as one table
            fetch = self._require_vector_fetch_fn()
            pair_off, ucount = lists.pair_off.cpu().numpy(), lists.ucount.cpu().numpy()
            cand = lists.cand_ids.cpu().numpy()
            rows_host = np.zeros(max(1, lists.total), dtype=np.int64)
            blocks, pos = [], 0
            for qi in np.flatnonzero(ucount):
                lo, u = int(pair_off[qi]), int(ucount[qi])
                blocks.append(self._fetch_checked(fetch, cand[lo:lo + u].tolist()))
                rows_host[lo:lo + u] = np.arange(pos, pos + u, dtype=np.int64)
                pos += u
            table = qd.upload(torch, np.concatenate(blocks, axis=0), dev)
            return qd.rank_and_cut(lists, top_k, top_p, queries_dev=x, corpus=table, cand_rows=qd.upload(torch, rows_host, dev))

    def _device_lists(self, qd, keys_dev, dev):
        """Every query's candidates, counted and ordered on the device: from the device mirror of the store's bucket arrays
        where the store keeps arrays, else from one ``get_bucket`` per (query, band) (the reference's storage interface,
        lshrs/storage/redis.py:282) with the pairs handed over flat."""
        st = self._storage
        if isinstance(st, _DeferredStorage):
            st = st._resolve()
        nq, nb, bb = (int(v) for v in keys_dev.shape)
        segs = st.array_segments(bb) if callable(getattr(st, "array_segments", None)) else None
        if segs is not None:
            desc, nseg, max_id = self._dev_buckets.table(segs, dev)
            return qd.candidates_from_index(keys_dev, desc, nseg, max_id)
        keys = keys_dev.cpu().numpy()
        ms, bs, off = [], [], np.zeros(nq + 1, dtype=np.int64)
        for qi in range(nq):
            n = 0
            for b in range(nb):
                mem = st.get_bucket(b, keys[qi, b].tobytes())
                if mem:
                    ms.append(np.fromiter((int(v) for v in mem), dtype=np.int64, count=len(mem)))
                    bs.append(np.full(len(mem), b, dtype=np.int32))
                    n += len(mem)
            off[qi + 1] = off[qi] + n
        members = np.concatenate(ms) if ms else np.empty(0, np.int64)
        bands = np.concatenate(bs) if bs else np.empty(0, np.int32)
        return qd.candidates_from_pairs(members, bands, off, nb, dev)

    def _fetch_checked(self, fetch, ids: list) -> np.ndarray:
        got = np.asarray(fetch(ids), dtype=np.float32)
        if got.ndim != 2 or got.shape[1] != self._dim:
            raise ValueError(f"Fetched vectors must have shape (n, {self._dim}); received {got.shape}")
        if got.shape[0] != len(ids):
            raise ValueError("vector_fetch_fn returned mismatched batch size "
                             f"(expected {len(ids)}, received {got.shape[0]})")
        return got

    def _query_many_host(self, arr: np.ndarray, top_k, top_p, corpus):
        """``query_many`` with the collision counting in NumPy between the signature launch and the rerank launch (round 5;
        what hashers without ``hash_device`` and batches beyond the device path's limits take).  Same arrays out."""
        nq = arr.shape[0]
        keys, flags = self._hasher.hash_batch_packed(arr, return_row_flags=True)
        if (flags & 1).any():
            raise ValueError(_ZERO_MSG)
        um, bounds = self._ordered_candidates_arrays(keys)
        lens = np.diff(bounds)
        if top_p is None:
            keep = lens if top_k is None else np.minimum(lens, top_k)
            return um[_ragged_positions(bounds[:-1], keep)], None, np.r_[0, np.cumsum(keep)].astype(np.int64)

        # rerank every non-empty candidate list in one launch: a (q, c_max) index matrix padded with -1
        # (out-of-range entries score NaN, which the device sort places last)
        c_max = int(lens.max()) if nq else 0
        if c_max == 0:
            return np.empty(0, np.int64), np.empty(0, np.float32), np.zeros(nq + 1, np.int64)
        rows = np.repeat(np.arange(nq, dtype=np.int64), lens)
        cols = np.arange(um.shape[0], dtype=np.int64) - np.repeat(bounds[:-1], lens)
        cand_ids = np.full((nq, c_max), -1, dtype=np.int64)
        cand_ids[rows, cols] = um
        if corpus is None:
            fetch = self._require_vector_fetch_fn()
            blocks = [self._fetch_checked(fetch, um[bounds[qi]:bounds[qi + 1]].tolist()) for qi in np.flatnonzero(lens)]
            table = np.concatenate(blocks, axis=0)
            cand = np.full((nq, c_max), -1, dtype=np.int64)
            cand[rows, cols] = np.arange(um.shape[0], dtype=np.int64)     # row of `table` = position in the flat list
        else:
            table, cand = corpus, cand_ids
        order, scores = _rerank_padded(arr, table, cand)                   # (q, c_max): positions, descending scores
        keep = np.maximum(1, np.ceil(lens * top_p).astype(np.int64))       # (reference: main.py:652-657)
        keep[lens == 0] = 0
        if top_k is not None:
            keep = np.minimum(keep, top_k)
        krows = np.repeat(np.arange(nq, dtype=np.int64), keep)
        kcols = np.arange(int(keep.sum()), dtype=np.int64) - np.repeat(np.cumsum(keep) - keep, keep)
        return (cand_ids[krows, order[krows, kcols]], np.ascontiguousarray(scores[krows, kcols], dtype=np.float32),
                np.r_[0, np.cumsum(keep)].astype(np.int64))

    # ------------------------------------------------------------------ storage pass-throughs
    def delete(self, indices: Union[int, Sequence[int]]) -> None:
        """Remove ids from every bucket (reference: main.py:744-784)."""
        to_remove = [indices] if isinstance(indices, int) else [int(i) for i in indices]
        self._storage.remove_indices(to_remove)

    def clear(self) -> None:
        """Flush what is buffered, then drop every bucket (reference: main.py:786-796)."""
        self.flush()
        self._storage.clear()

    def stats(self) -> Dict[str, Any]:
        """Static configuration summary (reference: main.py:798-840)."""
        return {
            "dimension": self._dim, "num_perm": self._config["num_perm"], "num_bands": self._config["num_bands"],
            "rows_per_band": self._config["rows_per_band"], "buffer_size": self._buffer_size,
            "similarity_threshold": self._config["similarity_threshold"], "redis_prefix": self._redis_config["prefix"],
        }

    # ------------------------------------------------------------------ persistence (same on-disk format)
    def save_to_disk(self, path) -> None:
        """``metadata.json`` (version, config, redis config with the password redacted) + ``projections.npz``
        (``arr_0 .. arr_{bands-1}``), the reference's format (main.py:846-895): an index saved by either
        implementation loads in the other.  Bucket contents live in the storage backend, not here."""
        self.flush()
        out = Path(path)
        out.mkdir(parents=True, exist_ok=True)
        redis_cfg = self._redis_config.copy()
        if "password" in redis_cfg:
            redis_cfg["password"] = "<REDACTED>"
        with open(out / "metadata.json", "w") as fh:
            meta = {"version": _FORMAT_VERSION, "config": self._config, "redis_config": redis_cfg}
            # (a key of our own: the reference's load_from_disk reads the three above only.)  Which BLAS the keys are the
            # reference's keys ON: the named build, or - for "host" - the build this host's NumPy was recognised as, so that a
            # loader on the other kind of host can tell (`_check_recorded_blas`)
            meta["lshrs_amd"] = self._blas_record()
            json.dump(meta, fh, indent=2)
        np.savez_compressed(out / "projections.npz", *self._hasher.projections)

    @classmethod
    def load_from_disk(cls, path, *, redis_config: Optional[Dict[str, Any]] = None,
                       vector_fetch_fn: Optional[VectorFetchFn] = None, storage: Any = None,
                       reference_blas: Optional[str] = None, strict_blas: Optional[bool] = None) -> "LSHRS":
        """Rebuild from :meth:`save_to_disk` output (reference: main.py:898-983).  The stored hyperplanes
        replace the freshly drawn ones — assigning ``_hasher.projections`` re-uploads the device image.
        ``reference_blas``: override the stored choice (e.g. the build a ``"host"`` index was recorded on);
        ``strict_blas``: raise instead of warning when a ``"host"`` index is loaded on a host whose BLAS sums its shape
        differently (default: the class attribute ``LSHRS.strict_blas``)."""
        src = Path(path)
        if not src.exists():
            raise FileNotFoundError(f"Directory not found: {src}")
        with open(src / "metadata.json") as fh:
            meta = json.load(fh)
        cfg = meta["config"]
        redis_cfg = meta["redis_config"].copy()
        if redis_config:
            redis_cfg.update(redis_config)
        extra = meta.get("lshrs_amd", {})
        blas = extra.get("reference_blas", "host") if reference_blas is None else reference_blas
        blas = _hostblas.LEGACY_BUILD_NAMES.get(blas, blas)
        _check_recorded_blas(blas, extra.get("host_blas"), cfg["rows_per_band"], cfg["dim"],
                             cls.strict_blas if strict_blas is None else strict_blas)
        inst = cls(
            dim=cfg["dim"], num_perm=cfg["num_perm"], num_bands=cfg["num_bands"], rows_per_band=cfg["rows_per_band"],
            similarity_threshold=cfg["similarity_threshold"], buffer_size=cfg["buffer_size"],
            vector_fetch_fn=vector_fetch_fn, storage=storage, redis_host=redis_cfg["host"], redis_port=redis_cfg["port"],
            redis_db=redis_cfg["db"], redis_password=redis_cfg["password"], redis_prefix=redis_cfg["prefix"],
            decode_responses=redis_cfg["decode_responses"], seed=cfg["seed"], reference_blas=blas)
        with np.load(src / "projections.npz") as data:
            inst._hasher.projections = [data[f"arr_{i}"].astype(np.float32) for i in range(len(data.files))]
        return inst

    def __getstate__(self) -> Dict[str, Any]:
        """Config + hyperplanes; buffer is flushed, fetch function and storage are not carried
        (reference: main.py:989-1008)."""
        self.flush()
        state = {"config": self._config.copy(), "redis_config": self._redis_config.copy(),
                 "projections": [np.asarray(m, dtype=np.float32) for m in self._hasher.projections]}
        # extras the reference's __setstate__ ignores (it reads the three keys above): what this build needs to come
        # back on the same device with the same windows and ingest mode
        h = self._hasher
        state["lshrs_amd"] = {
            "host_blas": self._blas_record().get("host_blas"),
            "packed_ingest": self._packed_ingest,
            "device": getattr(h, "_device", None) if isinstance(getattr(h, "_device", None), (int, str, type(None))) else str(h._device),
            "hasher_kwargs": {**{k: getattr(h, k) for k in ("tie_break", "precision", "tie_replay", "margin_guard",
                                                            "tie_threads", "audit_every", "audit_unflagged",
                                                            "reference_blas") if hasattr(h, k)},
                              **({"devices": list(h._devices)} if getattr(h, "_devices", None) else {})},
            "windows": {"tau_ulps": "bound" if getattr(h, "window_mode", {}).get("tau") == "bound" else getattr(h, "tau_ulps", 8.0),
                        "tau1_ulps": "bound" if getattr(h, "window_mode", {}).get("tau1") == "bound" else getattr(h, "tau1_ulps", 64.0)},
        }
        return state

    def __setstate__(self, state: Dict[str, Any]) -> None:
        cfg, rc = state["config"], state["redis_config"]
        extra = state.get("lshrs_amd", {})
        hasher = None
        if extra:
            hk = dict(extra.get("hasher_kwargs", {}))
            hk.pop("pipeline", None)               # (an option of earlier rounds: the interpreter-driven chunking is gone)
            if "reference_blas" in hk:
                hk["reference_blas"] = _hostblas.LEGACY_BUILD_NAMES.get(hk["reference_blas"], hk["reference_blas"])
            _check_recorded_blas(hk.get("reference_blas", "host"), extra.get("host_blas"), cfg["rows_per_band"], cfg["dim"],
                                 type(self).strict_blas)
            hk.update(extra.get("windows", {}))
            hasher = LSHHasher(num_bands=cfg["num_bands"], rows_per_band=cfg["rows_per_band"], dim=cfg["dim"],
                               seed=cfg["seed"], device=extra.get("device"), **hk)
        restored = self.__class__(
            dim=cfg["dim"], num_perm=cfg["num_perm"], num_bands=cfg["num_bands"], rows_per_band=cfg["rows_per_band"],
            similarity_threshold=cfg["similarity_threshold"], buffer_size=cfg["buffer_size"], vector_fetch_fn=None,
            redis_host=rc["host"], redis_port=rc["port"], redis_db=rc["db"], redis_password=rc["password"],
            redis_prefix=rc["prefix"], decode_responses=rc["decode_responses"], seed=cfg["seed"], hasher=hasher,
            packed_ingest=extra.get("packed_ingest", "auto"), storage=_DeferredStorage(rc))
        self.__dict__ = restored.__dict__
        self._hasher.projections = [np.asarray(m, dtype=np.float32) for m in state["projections"]]

    # ------------------------------------------------------------------ helpers
    strict_blas = False      # True: loading / unpickling a "host" index on a host of the other BLAS build raises (else: warns)

    def _blas_record(self) -> Dict[str, Any]:
        """What persistence writes about the BLAS the keys are the reference's keys on (`_check_recorded_blas` reads it)."""
        blas = getattr(self._hasher, "reference_blas", "host")
        rec: Dict[str, Any] = {"reference_blas": blas}
        if blas == "host":
            try:
                rec["host_blas"] = _hostblas.host_build_name()
            except Exception:           # pragma: no cover - a NumPy without OpenBLAS, the host library not built
                rec["host_blas"] = None
        return rec

    def _check_dim(self, vector) -> np.ndarray:
        """float32, flattened, right length (reference: main.py:1075-1080; the near-zero test of
        :1083 is evaluated by the kernel's row flag, see ``_ZERO_MSG`` call sites)."""
        resident = _device_tensor(vector)
        if resident is not None:       # ONE vector that lives on a GPU: the single-vector calls read theirs from (pinned) host memory
            vector = resident.detach().reshape(-1).cpu().numpy()
        arr = np.asarray(vector, dtype=np.float32).reshape(-1)
        if arr.shape[0] != self._dim:
            raise ValueError(f"Vector must have dimension {self._dim}; received {arr.shape[0]}")
        return arr

    def _hash_one(self, arr: np.ndarray):
        """(keys (bands, B), row flag) of one vector; concurrent callers share a launch where the hasher can coalesce."""
        one = getattr(self._hasher, "hash_one_packed", None)
        if one is not None:
            return one(arr)
        keys, flags = self._hasher.hash_batch_packed(arr.reshape(1, -1), return_row_flags=True)
        return keys[0], int(flags[0])

    def _candidate_counts_from_keys(self, band_keys: np.ndarray) -> Dict[int, int]:
        """Collision count per stored id over the query's band buckets (reference: main.py:1088-1111)."""
        counts: Dict[int, int] = {}
        for band_id in range(band_keys.shape[0]):
            for candidate in self._storage.get_bucket(band_id, band_keys[band_id].tobytes()):
                counts[candidate] = counts.get(candidate, 0) + 1
        return counts

    def _ordered_candidates_arrays(self, keys: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """For every query: the stored ids that share at least one band bucket with it, ordered by (-collisions, id) -
        ``_candidate_counts`` + the sort of ``query`` (lshrs/core/main.py:1088-1111, :614) for a whole batch, as array
        work: bucket members are gathered as flat (query, member) pairs, one sort counts the collisions, one orders the
        candidates.  Returns ``(ids, bounds)``: query ``i``'s candidates are ``ids[bounds[i]:bounds[i + 1]]``.  No Python
        object per member."""
        nq, nb = keys.shape[0], keys.shape[1]
        if hasattr(self._storage, "get_buckets_many"):
            q, m = self._storage.get_buckets_many(keys)
        else:   # the reference's storage interface: one bucket read per (query, band), members concatenated
            qs, ms = [], []
            for qi in range(nq):
                for band_id in range(nb):
                    mem = self._storage.get_bucket(band_id, keys[qi, band_id].tobytes())
                    if mem:
                        ms.append(np.fromiter((int(v) for v in mem), dtype=np.int64, count=len(mem)))
                        qs.append(np.full(len(mem), qi, dtype=np.int64))
            q = np.concatenate(qs) if qs else np.empty(0, np.int64)
            m = np.concatenate(ms) if ms else np.empty(0, np.int64)
        if q.size == 0:
            return np.empty(0, np.int64), np.zeros(nq + 1, dtype=np.int64)
        qbits, cbits = max(1, int(nq - 1).bit_length()), int(nb).bit_length()
        mbits = 63 - qbits - cbits
        if int(m.min()) >= 0 and int(m.max()) < (1 << mbits):
            # one 64-bit key per pair: (query, member), then (query, bands - collisions, member): two plain sorts
            pair = np.sort((q << mbits) | m)
            first = np.r_[True, pair[1:] != pair[:-1]]
            starts = np.flatnonzero(first)
            counts = np.diff(np.r_[starts, pair.shape[0]])     # (one pair per (query, band, member): the lookup's contract)
            uniq = pair[starts]
            uq, um = uniq >> mbits, uniq & ((1 << mbits) - 1)
            ranked = np.sort((uq << (mbits + cbits)) | ((nb - counts) << mbits) | um)
            uq, um = ranked >> (mbits + cbits), ranked & ((1 << mbits) - 1)
        else:
            order = np.lexsort((m, q))                           # by query, then member
            q, m = q[order], m[order]
            first = np.r_[True, (q[1:] != q[:-1]) | (m[1:] != m[:-1])]
            starts = np.flatnonzero(first)
            counts = np.diff(np.r_[starts, q.shape[0]])
            uq, um = q[starts], m[starts]
            rank = np.lexsort((um, -counts, uq))                 # by query, then -collisions, then id
            uq, um = uq[rank], um[rank]
        return um, np.searchsorted(uq, np.arange(nq + 1)).astype(np.int64)

    def _ordered_candidates_many(self, keys: np.ndarray) -> List[List[int]]:
        um, bounds = self._ordered_candidates_arrays(keys)
        return _split_rows(um.tolist(), np.diff(bounds))

    lane_rows = 262_144      # rows per unit when ONE index() call is cut up for several devices (two stream chunks)

    def _ingest_hashers(self) -> list:
        get = getattr(self._hasher, "device_hashers", None)
        return get() if callable(get) else [self._hasher]

    def _streams_buckets(self, n_ops: int) -> bool:
        """Does a batch of this many operations take the pipelined array path?  The storage must take bucket arrays
        (``batch_add_csr``, natively or through ``RedisPackedWriter``) and the hasher must be able to leave its keys on the
        device (``LSHHasher``; an injected hasher of another kind keeps the one-call path)."""
        sink = self._packed_sink(n_ops)
        return (sink is not None and hasattr(sink, "batch_add_csr")
                and callable(getattr(self._hasher, "device_hashers", None))
                and getattr(self._hasher, "tie_break", "host") == "host")

    def _open_ingest(self, inline: bool = False):
        from ._ingest import CsrIngest

        return CsrIngest(self._ingest_hashers(), self._packed_sink(1 << 62), lambda: ValueError(_ZERO_MSG),
                         lambda: ValueError("index must be non-negative"), inline=inline)

    def _check_batch(self, indices, vectors):
        """(ids int64, rows float32 C-contiguous) of one batch, with the reference's shape errors (main.py:504-511); rows that
        live on a GPU (a torch tensor) stay where they are."""
        resident = _device_tensor(vectors)
        arr = resident if resident is not None else np.asarray(vectors, dtype=np.float32)
        shape = tuple(int(v) for v in arr.shape)
        if len(shape) != 2 or shape[1] != self._dim:
            raise ValueError(f"Vectors must have shape (n, {self._dim}); received {shape}")
        if shape[0] != len(indices):
            raise ValueError(
                "Number of vectors does not match number of indices "
                f"(received {shape[0]} vectors for {len(indices)} indices)")
        id_arr = np.asarray(indices)
        id_arr = id_arr.astype(np.int64) if id_arr.dtype.kind in "iuf" else np.array([int(i) for i in indices], dtype=np.int64)
        return id_arr, (arr if resident is not None else np.ascontiguousarray(arr))

    def _packed_sink(self, n_ops: int):
        """The object ``index()`` hands a batch's buckets to as arrays, or None for the reference's operation tuples
        (see ``packed_ingest`` in the class docstring)."""
        mode = self._packed_ingest
        if mode is False or (mode == "auto" and n_ops < self.packed_auto_min_ops):
            return None
        st = self._storage
        if isinstance(st, _DeferredStorage):
            st = st._resolve()                   # (index() is about to write to it anyway)
        if hasattr(st, "batch_add_csr") or hasattr(st, "batch_add_packed"):
            return st
        if callable(getattr(st, "pipeline", None)) and callable(getattr(st, "bucket_key", None)):
            # the reference's RedisStorage (lshrs/storage/redis.py:187,508): same SADDs, one command per bucket
            if self._packed_writer is None or self._packed_writer.storage is not st:
                from .packed_ops import RedisPackedWriter

                self._packed_writer = RedisPackedWriter(st, flush_members=self._buffer_size)
            return self._packed_writer
        return None

    def _enqueue_packed(self, index: int, band_keys: np.ndarray) -> None:
        ops = [(b, band_keys[b].tobytes(), index) for b in range(band_keys.shape[0])]
        with self._buffer_lock:
            self._buffer.extend(ops)

    def _flush_buffer_if_needed(self) -> None:
        with self._buffer_lock:
            full = len(self._buffer) >= self._buffer_size
        if full:
            self.flush()

    def _require_vector_fetch_fn(self) -> VectorFetchFn:
        if self._vector_fetch_fn is None:
            raise RuntimeError("vector_fetch_fn must be supplied for operations requiring reranking")
        return self._vector_fetch_fn

    def _resolve_loader(self, format: str) -> Loader:
        """Loader lookup (reference: main.py:1159-1196).  The PostgreSQL / Parquet readers are the
        reference's own modules (I/O is out of scope here) and are imported from that package."""
        normalized = format.lower()
        if normalized in {"batches", "iter", "iterable"}:
            def _from_iterable(batches: Iterable[Tuple[Sequence[int], np.ndarray]]):
                yield from batches
            return _from_iterable
        if normalized in {"postgres", "pg"}:
            from lshrs.io.postgres import iter_postgres_vectors  # reference component
            return iter_postgres_vectors
        if normalized in {"parquet", "pq"}:
            from .parquet_fast import iter_parquet_vectors  # same contract as the reference's, array-based
            return iter_parquet_vectors
        raise ValueError(f"Unsupported signature creation format '{format}'")


lshrs = LSHRS  # lower-case alias kept by the reference (lshrs/core/main.py, last line)
