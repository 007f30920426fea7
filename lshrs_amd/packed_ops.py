"""Packed-key -> storage-operation path.

Once hashing runs at hundreds of millions of vectors per second, turning the ``(n, bands, B)`` key array
into ``n x bands`` Python ``(band, bytes, id)`` tuples (lshrs/core/main.py:1113-1128) and one ``SADD`` per
tuple (lshrs/storage/redis.py:408-416) is what bounds ingestion.  This module keeps the wire format — the
same bucket key text ``{prefix}:{band}:bucket:{hex}``, the same members — but works on whole arrays:

  * ``hex_keys``          all N x bands key texts in one device pass (``lshrs_keys_to_hex_u8``);
  * ``group_by_bucket``   per band, the distinct keys and the ids that fall into each (NumPy, no Python loop
                          over vectors);
  * ``bucket_csr``        the same grouping as ONE compressed-row structure for the whole batch - distinct buckets,
                          their offsets, the member ids bucket by bucket - built by a counting sort per band on the
                          device (``lshrs_bucket_histogram_u8`` / ``lshrs_bucket_scatter_u8``; keys of 1 or 2 bytes)
                          or by NumPy sorts (wider keys): no Python object per vector, per operation or per bucket;
  * ``RedisPackedWriter`` one ``SADD key m1 m2 ...`` per *bucket* through the reference's own
                          ``RedisStorage.pipeline()`` / ``bucket_key()`` (same set contents as one SADD per
                          member, far fewer commands);
  * ``InMemoryStorage.batch_add_packed`` / ``batch_add_csr`` (in storage.py) consume the same groups.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Iterator, List, Sequence, Tuple

import numpy as np

from . import _native

__all__ = ["hex_keys", "hex_keys_device", "group_by_bucket", "bucket_csr", "BucketCSR", "dedupe_csr", "RedisPackedWriter",
           "DeviceCSRJob"]


@dataclass
class BucketCSR:
    """The buckets one batch touches.  Bucket ``g`` is ``(bands[g], key_bytes[g].tobytes())`` and holds
    ``members[offsets[g]:offsets[g + 1]]`` (order inside a bucket unspecified: buckets are sets).  Buckets are sorted
    by ``codes`` = band << (8 * band_bytes) | little-endian key, which is what lookups bisect on."""

    band_bytes: int
    bands: np.ndarray        # (m,) int32
    key_bytes: np.ndarray    # (m, band_bytes) uint8
    codes: np.ndarray        # (m,) int64 for band_bytes <= 6, else None
    offsets: np.ndarray      # (m + 1,) int64
    members: np.ndarray      # (n * num_bands,) int64
    vectors: int
    distinct: bool = False   # True once no bucket lists a member twice (buckets are SETS: lshrs/storage/redis.py:408-416 SADD)

    def __len__(self) -> int:
        return int(self.bands.shape[0])


def ids_are_distinct(id_arr: np.ndarray) -> bool:
    """No id twice in a batch?  One pass when the ids ascend (the usual case), one sort otherwise."""
    if id_arr.shape[0] < 2 or bool((id_arr[1:] > id_arr[:-1]).all()):
        return True
    return int(np.unique(id_arr).shape[0]) == int(id_arr.shape[0])


def dedupe_csr(csr: "BucketCSR") -> "BucketCSR":
    """Set semantics for an array bucket list: every member at most once per bucket (the reference's buckets are Redis
    sets - an id indexed twice collides once per band, lshrs/core/main.py:1101-1109).  One sort of (bucket, member)."""
    if csr.distinct or len(csr) == 0:
        csr.distinct = True
        return csr
    lens = np.diff(csr.offsets)
    bucket = np.repeat(np.arange(len(csr), dtype=np.int64), lens)
    order = np.lexsort((csr.members, bucket))
    b, m = bucket[order], csr.members[order]
    keep = np.r_[True, (b[1:] != b[:-1]) | (m[1:] != m[:-1])]
    if keep.all():
        csr.distinct = True
        return csr
    b, m = b[keep], m[keep]
    new_lens = np.bincount(b, minlength=len(csr)).astype(np.int64)
    return BucketCSR(csr.band_bytes, csr.bands, csr.key_bytes, csr.codes, np.r_[0, np.cumsum(new_lens)].astype(np.int64),
                     m, csr.vectors, True)


def merge_csr(segments) -> "BucketCSR":
    """Several :class:`BucketCSR` with codes (``band_bytes <= 6``, all equal) folded into one: the union of their buckets,
    every bucket's members concatenated in segment order, then each member once per bucket (an id indexed in two calls
    is in its bucket ONCE: set semantics).  Array work only."""
    segs = [s for s in segments if len(s)]
    if not segs:
        return segments[0]
    bb = segs[0].band_bytes
    codes = np.concatenate([s.codes for s in segs])
    lens = np.concatenate([np.diff(s.offsets) for s in segs])
    uniq, inv = np.unique(codes, return_inverse=True)
    order = np.argsort(inv, kind="stable")                   # old buckets by new bucket, segment order inside
    new_lens = np.bincount(inv, weights=lens, minlength=uniq.shape[0]).astype(np.int64)
    new_off = np.r_[0, np.cumsum(new_lens)].astype(np.int64)
    # where every old bucket's members go: the running sum of the lengths of the old buckets sorted before it
    sorted_lens = lens[order]
    dst_start = np.empty(lens.shape[0], dtype=np.int64)
    dst_start[order] = np.cumsum(sorted_lens) - sorted_lens  # (laid out in exactly that order)
    members = np.empty(int(new_off[-1]), dtype=np.int64)
    src = np.concatenate([s.members for s in segs])
    total = int(lens.sum())
    pos = np.arange(total, dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens) + np.repeat(dst_start, lens)
    members[pos] = src
    kb = np.empty((uniq.shape[0], bb), dtype=np.uint8)
    for j in range(bb):
        kb[:, j] = (uniq >> (8 * j)) & 0xFF
    return dedupe_csr(BucketCSR(bb, (uniq >> (8 * bb)).astype(np.int32), kb, uniq, new_off, members,
                                sum(int(s.vectors) for s in segs)))


def key_codes(keys: np.ndarray) -> np.ndarray:
    """(..., bands, B) key bytes -> (..., bands) int64 codes ``band << 8B | little-endian key`` (B <= 6)."""
    keys = np.ascontiguousarray(keys, dtype=np.uint8)
    nb, bb = keys.shape[-2], keys.shape[-1]
    if bb > 6:
        raise ValueError("key codes need band keys of at most 6 bytes")
    code = np.zeros(keys.shape[:-1], dtype=np.int64)
    for j in range(bb):
        code |= keys[..., j].astype(np.int64) << (8 * j)
    return code | (np.arange(nb, dtype=np.int64) << (8 * bb))


def _csr_host(id_arr: np.ndarray, keys: np.ndarray) -> BucketCSR:
    n, nb, bb = keys.shape
    if n == 0:
        return BucketCSR(bb, np.empty(0, np.int32), np.empty((0, bb), np.uint8), np.empty(0, np.int64) if bb <= 6 else None,
                         np.zeros(1, np.int64), np.empty(0, np.int64), 0)
    if bb <= 6:
        codes = key_codes(keys).reshape(-1)                      # vector-major, band-minor
        order = np.argsort(codes, kind="stable")
        sc = codes[order]
        starts = np.flatnonzero(np.r_[True, sc[1:] != sc[:-1]])
        ucodes = sc[starts]
        bands = (ucodes >> (8 * bb)).astype(np.int32)
        kb = np.empty((ucodes.shape[0], bb), dtype=np.uint8)
        for j in range(bb):
            kb[:, j] = (ucodes >> (8 * j)) & 0xFF
        members = np.repeat(id_arr, nb)[order]
        return BucketCSR(bb, bands, kb, ucodes, np.r_[starts, n * nb].astype(np.int64), members, n)
    parts = [(band, key, mem) for band, key, mem in group_by_bucket(id_arr, keys)]
    bands = np.array([p[0] for p in parts], dtype=np.int32)
    kb = np.frombuffer(b"".join(p[1] for p in parts), dtype=np.uint8).reshape(-1, bb).copy()
    lens = np.array([len(p[2]) for p in parts], dtype=np.int64)
    return BucketCSR(bb, bands, kb, None, np.r_[0, np.cumsum(lens)].astype(np.int64),
                     np.concatenate([p[2] for p in parts]) if parts else np.empty(0, np.int64), n)


def _csr_device_sorted(torch, id_arr: np.ndarray, keys, device) -> BucketCSR:
    """Band keys of 3 to 6 bytes (config 5: 4) do not fit a counting-sort table: their (band, key) codes are sorted on the
    device instead (one stable 64-bit sort of n x bands codes through the runtime's radix sort - data movement, no
    arithmetic of the path in it) and run-length encoded; same structure, same bucket order as the other two builders."""
    on_dev = isinstance(keys, torch.Tensor)
    if on_dev:
        kd = keys.contiguous()
        dev = kd.device
    else:
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        kd = torch.from_numpy(np.ascontiguousarray(keys, dtype=np.uint8)).to(dev)
    n, nb, bb = (int(v) for v in kd.shape)
    with torch.cuda.device(dev):
        codes = torch.zeros((n, nb), dtype=torch.int64, device=dev)
        for j in range(bb):
            codes |= kd[:, :, j].to(torch.int64) << (8 * j)
        codes |= (torch.arange(nb, dtype=torch.int64, device=dev) << (8 * bb))[None, :]
        sorted_codes, order = torch.sort(codes.reshape(-1), stable=True)
        members = torch.from_numpy(id_arr).to(dev).repeat_interleave(nb)[order]
        ucodes, counts = torch.unique_consecutive(sorted_codes, return_counts=True)
        ucodes_h, counts_h, members_h = ucodes.cpu().numpy(), counts.cpu().numpy(), members.cpu().numpy()
    kb = np.empty((ucodes_h.shape[0], bb), dtype=np.uint8)
    for j in range(bb):
        kb[:, j] = (ucodes_h >> (8 * j)) & 0xFF
    return BucketCSR(bb, (ucodes_h >> (8 * bb)).astype(np.int32), kb, ucodes_h,
                     np.r_[0, np.cumsum(counts_h, dtype=np.int64)].astype(np.int64), members_h, n)


def bucket_csr(ids: Sequence[int], keys, *, device=None) -> BucketCSR:
    """Group a batch's ``(n, bands, B)`` keys into buckets: one :class:`BucketCSR`.  ``B <= 2`` (every BASELINE config
    but config 5): counting sort on the device (``lshrs_bucket_histogram_u8`` / ``_scatter_u8``); ``B`` 3 to 6 (config 5):
    a device sort of the (band, key) codes; wider keys: NumPy sorts on the host.  ``keys`` may be the NumPy array
    ``hash_batch_packed`` returns or the device tensor ``hash_device`` returns."""
    torch = _native.require_gpu()
    id_arr = np.ascontiguousarray(np.asarray(ids, dtype=np.int64))
    on_dev = isinstance(keys, torch.Tensor)
    n, nb, bb = (int(v) for v in keys.shape)
    if id_arr.shape[0] != n:
        raise ValueError("ids and keys disagree in length")
    if n == 0 or bb > 6 or bb > 2:
        if n == 0 or bb > 6:
            host_keys = keys.cpu().numpy() if on_dev else np.ascontiguousarray(keys, dtype=np.uint8)
            csr = _csr_host(id_arr, host_keys)
        else:
            csr = _csr_device_sorted(torch, id_arr, keys, device)
        csr.distinct = ids_are_distinct(id_arr)
        return csr if csr.distinct else dedupe_csr(csr)
    lib = _native.load()
    if on_dev:
        kd = keys.contiguous()
        dev = kd.device
    else:
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        kd = torch.from_numpy(np.ascontiguousarray(keys, dtype=np.uint8)).to(dev)
    bins = nb << (8 * bb)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        idd = torch.from_numpy(id_arr).to(dev)
        counts = torch.zeros(bins, dtype=torch.int32, device=dev)
        _native.check(lib.lshrs_bucket_histogram_u8(kd.data_ptr(), n, nb, bb, counts.data_ptr(), stream),
                      "lshrs_bucket_histogram_u8")
        ends = torch.cumsum(counts, 0, dtype=torch.int64)
        offsets = ends - counts
        cursors = torch.zeros(bins, dtype=torch.int32, device=dev)
        members = torch.empty(n * nb, dtype=torch.int64, device=dev)
        _native.check(lib.lshrs_bucket_scatter_u8(kd.data_ptr(), idd.data_ptr(), n, nb, bb, offsets.data_ptr(),
                                                  cursors.data_ptr(), members.data_ptr(), stream),
                      "lshrs_bucket_scatter_u8")
        counts_h = counts.cpu().numpy()
        members_h = members.cpu().numpy()
    live = np.flatnonzero(counts_h).astype(np.int64)             # ascending bin = ascending code
    kb = np.empty((live.shape[0], bb), dtype=np.uint8)
    for j in range(bb):
        kb[:, j] = (live >> (8 * j)) & 0xFF
    csr = BucketCSR(bb, (live >> (8 * bb)).astype(np.int32), kb, live,
                    np.r_[0, np.cumsum(counts_h[live], dtype=np.int64)].astype(np.int64), members_h, n,
                    ids_are_distinct(id_arr))
    return csr if csr.distinct else dedupe_csr(csr)      # (an id twice in one batch: once per bucket, as SADD leaves it)


_COPY_POOL = None


def _copy_pool():
    """Six helper threads for the large host copies of :meth:`DeviceCSRJob.finish` (created at first use, per process)."""
    global _COPY_POOL
    if _COPY_POOL is None or getattr(_COPY_POOL, "_pid", None) != __import__("os").getpid():
        from concurrent.futures import ThreadPoolExecutor

        _COPY_POOL = ThreadPoolExecutor(max_workers=6, thread_name_prefix="lshrs-copy")
        _COPY_POOL._pid = __import__("os").getpid()
    return _COPY_POOL


class DeviceCSRJob:
    """:func:`bucket_csr` for keys that are ALREADY on the device, in two halves (round 5, the streamed ingest of
    ``LSHRS.index``): the constructor only ENQUEUES - ids to the device, histogram, scan, scatter, the counts and the
    members into pinned host blocks - on the CURRENT stream and returns; :meth:`finish` (any thread, once the stream has
    reached ``event``) turns the pinned blocks into the :class:`BucketCSR` of arrays the stores take.  Between the two
    the caller streams the next chunk: the grouping of chunk i rides under the host->device copy of chunk i + 1.
    Keys of 1 or 2 bytes (the counting sort); wider keys: ``bucket_csr`` itself (one device sort, synchronous)."""

    def __init__(self, ids: np.ndarray, keys_dev, ids_dev=None) -> None:
        torch = _native.require_gpu()
        lib = _native.load()
        self.ids = np.ascontiguousarray(ids, dtype=np.int64)
        n, nb, bb = (int(v) for v in keys_dev.shape)
        if self.ids.shape[0] != n:
            raise ValueError("ids and keys disagree in length")
        if bb > 2 or n == 0:
            raise ValueError("DeviceCSRJob takes non-empty batches with band keys of 1 or 2 bytes")
        self.n, self.nb, self.bb = n, nb, bb
        dev = keys_dev.device
        bins = nb << (8 * bb)
        cur = torch.cuda.current_stream(dev)
        stream = cur.cuda_stream
        kd = keys_dev if keys_dev.is_contiguous() else keys_dev.contiguous()
        # (`ids_dev`: the same ids already on the device - a caller that streams chunks uploads a batch's ids ONCE, in front of
        #  the stream: a small host->device copy issued between the chunks queues behind the next chunk's 400 MB on the copy
        #  engine and the grouping waits 6 ms for 1 MB)
        idd = ids_dev if ids_dev is not None else torch.from_numpy(self.ids).to(dev, non_blocking=True)
        counts = torch.zeros(bins, dtype=torch.int32, device=dev)
        _native.check(lib.lshrs_bucket_histogram_u8(kd.data_ptr(), n, nb, bb, counts.data_ptr(), stream),
                      "lshrs_bucket_histogram_u8")
        ends = torch.cumsum(counts, 0, dtype=torch.int64)
        offsets = ends - counts
        cursors = torch.zeros(bins, dtype=torch.int32, device=dev)
        members = torch.empty(n * nb, dtype=torch.int64, device=dev)
        _native.check(lib.lshrs_bucket_scatter_u8(kd.data_ptr(), idd.data_ptr(), n, nb, bb, offsets.data_ptr(),
                                                  cursors.data_ptr(), members.data_ptr(), stream),
                      "lshrs_bucket_scatter_u8")
        # The live buckets - bins with at least one member - compacted ON THE DEVICE with static shapes (no size comes back to
        # the host in between): position of every live bin by a scan of the mask, one scatter through it (dead bins go to a
        # spare slot), then code, band, key bytes and end offset of every live bucket as dense arrays.  NumPy needs 6-9 ms per
        # 2^20 bins for the same (flatnonzero, fancy indexing, a strided byte gather); here it rides under the next chunk's copy.
        mask = counts > 0
        pos = torch.cumsum(mask, 0, dtype=torch.int64) - 1
        slot = torch.where(mask, pos, torch.full_like(pos, bins))
        live = torch.empty(bins + 1, dtype=torch.int64, device=dev)
        live.scatter_(0, slot, torch.arange(bins, dtype=torch.int64, device=dev))
        live = live[:bins]
        n_live = pos[-1:] + 1
        safe = live.clamp_(0, bins - 1)                   # (entries behind n_live are whatever was there: never read)
        # (what crosses the link is as narrow as it can be: codes and end offsets as int32 where they fit - bins and n * bands
        #  below 2^31, every BASELINE config -, no bands: those are the codes' high bits)
        narrow = bins < (1 << 31) and n * nb < (1 << 31)
        off_end = ends.index_select(0, safe)
        shifts = torch.arange(bb, dtype=torch.int64, device=dev) * 8
        kb = ((safe[:, None] >> shifts[None, :]) & 0xFF).to(torch.uint8)
        live_out, off_out = (safe.to(torch.int32), off_end.to(torch.int32)) if narrow else (safe, off_end)
        # (pinned blocks from torch's caching host allocator: after the first batches these are reused, not allocated)
        pin = lambda t: torch.empty(t.shape, dtype=t.dtype, pin_memory=True)      # noqa: E731
        self._host = [pin(n_live), pin(live_out), pin(off_out), pin(kb), pin(members)]
        for h, d in zip(self._host, (n_live, live_out, off_out, kb, members)):
            d = d.contiguous()
            # (by a kernel, not a copy engine: a memcpy issued here waits every third time for the next chunk's 400 MB
            #  host->device copy to finish - include/lshrs_hip.h, lshrs_copy_to_host_u8)
            _native.check(lib.lshrs_copy_to_host_u8(d.data_ptr(), h.data_ptr(), d.numel() * d.element_size(), stream),
                          "lshrs_copy_to_host_u8")
            d.record_stream(cur)
        # what a query's bucket lookup needs of this grouping stays ON THE DEVICE (round 6, `_query_device.DeviceBuckets`): the live
        # codes, the offsets with a leading zero (static shape: entry g + 1 = the end of live bucket g; what lies behind the
        # live ones is never read) and the members - the store's segment carries them, so the first query after an ingest does
        # not upload what was here a moment ago (240 MB per 1 M x 16 bands)
        off_full = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), off_end])
        self._dev = (dev, safe, off_full, members)
        self.event = torch.cuda.Event()
        self.event.record(cur)
        for t in (kd, idd, counts, ends, offsets, cursors, members, mask, pos, slot, safe, off_end, kb, n_live, live_out, off_out, off_full):
            t.record_stream(cur)                          # (used on `cur`, perhaps allocated under another stream)

    def finish(self) -> BucketCSR:
        """Wait for the device half, then the host half: the first ``n_live`` entries of the dense arrays and the members,
        copied out of the pinned blocks (which go back to the allocator: a store keeps its arrays for as long as it lives,
        and page-locked memory is not what it should be keeping them in)."""
        self.event.synchronize()
        n_live_h, live_h, off_h, kb_h, members_h = (t.numpy() for t in self._host)
        m = int(n_live_h[0])
        # ~40 MB of copies and widenings per 131 072 x 16 operations: memory-bound, 4 ms on one thread.  NumPy releases the
        # interpreter lock inside each of them, so a few helper threads run them side by side (1.2-1.9 ms).  (torch's CPU copy from
        # a worker thread stalls in its own thread pool - 70 ms per chunk, measured - hence NumPy.)
        codes = np.empty(m, dtype=np.int64)
        offsets = np.empty(m + 1, dtype=np.int64)
        offsets[0] = 0
        members = np.empty(members_h.shape[0], dtype=np.int64)
        kb = np.empty((m, self.bb), dtype=np.uint8)
        bands = np.empty(m, dtype=np.int32)
        q = -(-members.shape[0] // 4)
        tasks = [(lambda a=a: np.copyto(members[a:a + q], members_h[a:a + q])) for a in range(0, members.shape[0], q)]
        tasks += [lambda: np.copyto(codes, live_h[:m], casting="unsafe"),
                  lambda: np.right_shift(live_h[:m], 8 * self.bb, out=bands, casting="unsafe"),
                  lambda: np.copyto(offsets[1:], off_h[:m], casting="unsafe"), lambda: np.copyto(kb, kb_h[:m])]
        if members.shape[0] >= (1 << 18):
            futures = [_copy_pool().submit(t) for t in tasks[:-1]]
            tasks[-1]()
            for f in futures:
                f.result()
        else:
            for t in tasks:
                t()
        csr = BucketCSR(self.bb, bands, kb, codes, offsets, members, self.n, ids_are_distinct(self.ids))
        self._host = None
        if csr.distinct:
            dev, codes_d, off_d, members_d = self._dev
            csr._dev = (dev, codes_d[:m], off_d[:m + 1], members_d)     # (views: the device copies of exactly these arrays)
            return csr
        return dedupe_csr(csr)        # (an id twice in the batch: the segment is rebuilt on the host - and uploaded when a query needs it)


def hex_keys_device(keys):
    """Device ``(n, bands, B)`` uint8 keys -> device ``(n, bands, 2B)`` uint8 ASCII (lower-case hex)."""
    torch = _native.require_gpu()
    lib = _native.load()
    if keys.dtype != torch.uint8 or not keys.is_cuda or keys.dim() != 3:
        raise TypeError("hex_keys_device expects a uint8 device tensor of shape (n, bands, band_bytes)")
    keys = keys.contiguous()
    out = torch.empty((keys.shape[0], keys.shape[1], 2 * keys.shape[2]), dtype=torch.uint8, device=keys.device)
    if keys.numel():
        with torch.cuda.device(keys.device):
            stream = torch.cuda.current_stream(keys.device).cuda_stream
            _native.check(lib.lshrs_keys_to_hex_u8(keys.data_ptr(), keys.numel(), out.data_ptr(), stream),
                          "lshrs_keys_to_hex_u8")
    return out


def hex_keys(keys) -> np.ndarray:
    """``(n, bands)`` array of ``bytes`` (dtype ``S{2B}``): element [i, b] == ``keys[i, b].tobytes().hex().encode()``.
    Accepts the NumPy key array ``LSHHasher.hash_batch_packed`` returns or a device tensor."""
    torch = _native.require_gpu()
    dev = keys if isinstance(keys, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(keys, dtype=np.uint8)).cuda()
    hx = hex_keys_device(dev).cpu().numpy()
    return np.ascontiguousarray(hx).view(f"S{hx.shape[2]}")[:, :, 0]


def _band_codes(col: np.ndarray) -> np.ndarray:
    """(n, B) key bytes of one band -> one sortable integer per row (B <= 8), else a void view."""
    n, bb = col.shape
    if bb <= 8:
        wide = np.zeros((n, 8), dtype=np.uint8)
        wide[:, :bb] = col
        return wide.view(np.uint64)[:, 0]
    return np.ascontiguousarray(col).view(np.dtype((np.void, bb)))[:, 0]


def group_by_bucket(ids: Sequence[int], keys: np.ndarray) -> Iterator[Tuple[int, bytes, np.ndarray]]:
    """Yield ``(band, key_bytes, member_ids)`` for every distinct bucket touched by this batch; members keep
    their order of appearance.  ``keys`` is the ``(n, bands, B)`` uint8 array; pure data movement (one stable
    integer sort per band, no Python loop over vectors)."""
    keys = np.ascontiguousarray(keys, dtype=np.uint8)
    id_arr = np.asarray(ids, dtype=np.int64)
    n, nb, bb = keys.shape
    if id_arr.shape[0] != n:
        raise ValueError("ids and keys disagree in length")
    if n == 0:
        return
    for band in range(nb):
        col = keys[:, band, :]
        codes = _band_codes(col)
        order = np.argsort(codes, kind="stable")
        sorted_codes = codes[order]
        starts = np.flatnonzero(np.r_[True, sorted_codes[1:] != sorted_codes[:-1]])
        stops = np.r_[starts[1:], n]
        members_sorted = id_arr[order]
        first_rows = order[starts]
        key_blob = np.ascontiguousarray(col[first_rows]).tobytes()
        for g in range(len(starts)):
            yield band, key_blob[g * bb:(g + 1) * bb], members_sorted[starts[g]:stops[g]]


class RedisPackedWriter:
    """``batch_add_packed`` / ``batch_add_csr`` for the reference's ``RedisStorage`` (or anything with its ``pipeline()`` context
    manager and ``bucket_key()``): one pipelined ``SADD key m1 m2 ...`` per bucket.  A pipeline is executed and a new one
    opened every ``flush_members`` set members (``LSHRS`` passes its ``buffer_size``): what the reference's ``index()`` does by
    flushing its operation buffer every ``buffer_size`` operations (lshrs/core/main.py:1131-1143, redis.py:348-416) - a
    1 M x 16-band batch is then ~1 600 pipelines of 10 000 members, not one of 16 M: bounded client memory, bounded
    blocking time, and a failure loses at most one pipeline's worth (``pipelines`` counts those executed)."""

    def __init__(self, storage, *, max_members_per_command: int = 4096, flush_members: int = 10_000) -> None:
        self.storage = storage
        self.max_members = int(max_members_per_command)
        self.flush_members = max(1, int(flush_members))
        self.pipelines = 0

    def _send(self, buckets) -> int:
        """``buckets`` yields (key text, member list); pipelines of at most ``flush_members`` members (a bucket larger than
        that is cut at command boundaries)."""
        commands = 0
        it = iter(buckets)
        pending = next(it, None)
        while pending is not None:
            room = self.flush_members
            with self.storage.pipeline() as pipe:
                self.pipelines += 1
                while pending is not None and room > 0:
                    name, members = pending
                    take = min(len(members), room, self.max_members)
                    pipe.sadd(name, *members[:take])
                    commands += 1
                    room -= take
                    pending = (name, members[take:]) if take < len(members) else next(it, None)
        return commands

    def batch_add_packed(self, ids: Sequence[int], keys: np.ndarray) -> int:
        return self._send((self.storage.bucket_key(band, key_bytes), [int(m) for m in members])
                          for band, key_bytes, members in group_by_bucket(ids, keys))

    def batch_add_csr(self, csr: BucketCSR) -> int:
        """The same commands from a :class:`BucketCSR`: key texts by ``bucket_key`` (the reference's format), members
        as array slices."""
        return self._send((self.storage.bucket_key(int(csr.bands[g]), csr.key_bytes[g].tobytes()),
                           csr.members[int(csr.offsets[g]):int(csr.offsets[g + 1])].tolist()) for g in range(len(csr)))

    def __getattr__(self, item):  # everything else (get_bucket, batch_add, close, ...) is the wrapped storage's
        return getattr(self.storage, item)
