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

__all__ = ["hex_keys", "hex_keys_device", "group_by_bucket", "bucket_csr", "BucketCSR", "dedupe_csr", "RedisPackedWriter"]


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
    """``batch_add_packed`` for the reference's ``RedisStorage`` (or anything with its ``pipeline()`` context
    manager and ``bucket_key()``): one pipelined ``SADD key m1 m2 ...`` per bucket."""

    def __init__(self, storage, *, max_members_per_command: int = 4096) -> None:
        self.storage = storage
        self.max_members = int(max_members_per_command)

    def batch_add_packed(self, ids: Sequence[int], keys: np.ndarray) -> int:
        commands = 0
        with self.storage.pipeline() as pipe:
            for band, key_bytes, members in group_by_bucket(ids, keys):
                name = self.storage.bucket_key(band, key_bytes)
                for lo in range(0, len(members), self.max_members):
                    pipe.sadd(name, *[int(m) for m in members[lo:lo + self.max_members]])
                    commands += 1
        return commands

    def batch_add_csr(self, csr: BucketCSR) -> int:
        """The same commands from a :class:`BucketCSR`: key texts by ``bucket_key`` (the reference's format), members
        as array slices."""
        commands = 0
        with self.storage.pipeline() as pipe:
            for g in range(len(csr)):
                name = self.storage.bucket_key(int(csr.bands[g]), csr.key_bytes[g].tobytes())
                lo, hi = int(csr.offsets[g]), int(csr.offsets[g + 1])
                for a in range(lo, hi, self.max_members):
                    pipe.sadd(name, *csr.members[a:min(hi, a + self.max_members)].tolist())
                    commands += 1
        return commands

    def __getattr__(self, item):  # everything else (get_bucket, batch_add, close, ...) is the wrapped storage's
        return getattr(self.storage, item)
