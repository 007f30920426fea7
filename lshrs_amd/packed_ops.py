"""Packed-key -> storage-operation path.

Once hashing runs at hundreds of millions of vectors per second, turning the ``(n, bands, B)`` key array
into ``n x bands`` Python ``(band, bytes, id)`` tuples (lshrs/core/main.py:1113-1128) and one ``SADD`` per
tuple (lshrs/storage/redis.py:408-416) is what bounds ingestion.  This module keeps the wire format — the
same bucket key text ``{prefix}:{band}:bucket:{hex}``, the same members — but works on whole arrays:

  * ``hex_keys``          all N x bands key texts in one device pass (``lshrs_keys_to_hex_u8``);
  * ``group_by_bucket``   per band, the distinct keys and the ids that fall into each (NumPy, no Python loop
                          over vectors);
  * ``RedisPackedWriter`` one ``SADD key m1 m2 ...`` per *bucket* through the reference's own
                          ``RedisStorage.pipeline()`` / ``bucket_key()`` (same set contents as one SADD per
                          member, far fewer commands);
  * ``InMemoryStorage.batch_add_packed`` (in storage.py) consumes the same groups.
"""

from __future__ import annotations

from typing import Iterator, List, Sequence, Tuple

import numpy as np

from . import _native

__all__ = ["hex_keys", "hex_keys_device", "group_by_bucket", "RedisPackedWriter"]


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

    def __getattr__(self, item):  # everything else (get_bucket, batch_add, close, ...) is the wrapped storage's
        return getattr(self.storage, item)
