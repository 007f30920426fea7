"""The candidate path of ``LSHRS.query_many`` on the device (round 6; SURVEY.md §8f row 2, the device variant).

Per query the reference reads one bucket per band, counts in a dict how often every stored id turns up, sorts by
``(-count, id)``, fetches the candidates' vectors, ranks them by cosine and cuts the list (lshrs/core/main.py:524-658,
``_candidate_counts`` :1088-1111).  Here a whole batch of queries does that between ONE upload (the query vectors) and ONE
download (ids, scores, bounds):

  signature pass -> ``lshrs_query_lookup_u8`` (bisection in the device-resident bucket arrays) -> scan ->
  ``lshrs_query_collide_*`` (sort / count / order inside a workgroup's LDS) -> [``lshrs_cosine_ragged_f32`` on the resident
  corpus] -> ``lshrs_query_rank_f32`` (order by score, cut to top-p / top-k, compact)

``DeviceBuckets`` mirrors the array segments of a store (``InMemoryStorage.array_segments``) in device memory: uploaded
the first time a query meets them, kept until the store replaces them.  Stores that only answer ``get_bucket`` (the
reference's ``RedisStorage``) hand their (query, member, band) pairs over flat (``collide_pairs``).  No CPU compute path:
the host only moves arrays.
"""

from __future__ import annotations

import threading
from typing import List, Optional, Tuple

import numpy as np

from . import _native
from .windows import _U

__all__ = ["DeviceBuckets", "TooLarge", "candidates_from_index", "candidates_from_pairs", "rank_and_cut"]


class TooLarge(Exception):
    """A query's lists do not fit the LDS-resident networks (``LSHRS_QUERY_MAX_PAIRS``) or an id does not fit the item
    layout: the caller counts this batch on the host."""


def upload(torch, a: np.ndarray, dev):
    """Host array -> tensor on ``dev`` (the array is only read: a read-only view goes up without a defensive copy)."""
    import warnings

    a = np.ascontiguousarray(a)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        return torch.from_numpy(a).to(dev)


def _bbits(num_bands: int) -> int:
    return max(0, int(num_bands - 1).bit_length())


class DeviceBuckets:
    """Device mirror of a store's array segments: per segment ``codes`` / ``offsets`` / ``members`` as int64 tensors and the
    descriptor table ``lshrs_bucket_segment[nseg]`` the kernels read.  ``table(segments, dev)`` is cheap while the store's
    segment list has not changed."""

    def __init__(self) -> None:
        self._lock = threading.Lock()
        self._views: dict = {}           # id(segment) -> (segment, device index, codes, offsets, members)
        self._table = None               # (key, descriptor tensor, views kept alive, largest id)
        self.uploads = 0
        self.upload_bytes = 0
        self.kept_from_ingest = 0        # segments whose device arrays the ingest left behind (no upload)
        self.directory_max_codes = 1 << 22      # code spaces up to this size get a dense directory (16 MB of int32 per segment)

    def clear(self) -> None:
        with self._lock:
            self._views, self._table = {}, None

    def table(self, segments: list, dev):
        """(descriptor tensor or None for an empty index, nseg, largest member id) for this segment list on ``dev``."""
        torch = _native.require_gpu()
        segs = [s for s in segments if len(s)]
        key = (dev.index, tuple(id(s) for s in segs))
        with self._lock:
            if self._table is not None and self._table[0] == key:
                return self._table[1], len(segs), self._table[3]
            views, max_id = {}, -1
            rows = []
            with torch.cuda.device(dev):
                for s in segs:
                    v = self._views.get(id(s))
                    if v is None or v[0] is not s or v[1] != dev.index:
                        span = (int(s.members.min()), int(s.members.max())) if s.members.size else (0, -1)
                        kept = getattr(s, "_dev", None)        # the ingest's own device arrays of this segment (DeviceCSRJob)
                        if kept is not None and kept[0] == dev and int(kept[1].numel()) == len(s):
                            v = (s, dev.index, kept[1], kept[2], kept[3], span)
                            self.kept_from_ingest += 1
                        else:
                            codes = torch.from_numpy(np.ascontiguousarray(s.codes, dtype=np.int64)).to(dev)
                            offsets = torch.from_numpy(np.ascontiguousarray(s.offsets, dtype=np.int64)).to(dev)
                            members = torch.from_numpy(np.ascontiguousarray(s.members, dtype=np.int64)).to(dev)
                            v = (s, dev.index, codes, offsets, members, span)
                            self.uploads += 1
                            self.upload_bytes += 8 * (codes.numel() + offsets.numel() + members.numel())
                    if len(v) == 6:
                        # keys of 1 or 2 bytes: every possible code listed - directory[c] = the first bucket with code >= c -, built
                        # once per segment on the device (one searchsorted); the lookup reads it instead of bisecting
                        directory, dir_codes = None, 0
                        space = int(s.bands.max() + 1) << (8 * s.band_bytes) if len(s) else 0
                        if s.band_bytes <= 2 and 0 < space <= self.directory_max_codes:
                            every = torch.arange(space + 1, dtype=torch.int64, device=dev)
                            directory = torch.searchsorted(v[2], every).to(torch.int32)
                            dir_codes = space
                        v = v + (directory, dir_codes)
                    views[id(s)] = v
                    if v[5][0] < 0:
                        raise TooLarge("negative member id")
                    max_id = max(max_id, v[5][1])
                    rows.append([v[2].data_ptr(), v[3].data_ptr(), v[4].data_ptr(), int(v[2].numel()),
                                 v[6].data_ptr() if v[6] is not None else 0, v[7]])
                desc = torch.tensor(rows, dtype=torch.int64).to(dev) if rows else None
                if rows:
                    torch.cuda.current_stream(dev).synchronize()
            self._views = views          # (segments the store dropped go with their device copies)
            self._table = (key, desc, views, max_id)
            return desc, len(segs), max_id


class _Lists:
    """What the collide step leaves on the device: query i's candidates are ``cand_ids[pair_off[i] : pair_off[i] + ucount[i]]``,
    ordered by (-collisions, id); ``hits`` their collision counts."""

    __slots__ = ("nq", "total", "max_pairs", "pair_off", "cand_ids", "hits", "ucount", "dev", "segments", "big")

    def __init__(self, nq, total, max_pairs, pair_off, cand_ids, hits, ucount, dev, segments=0, big=()):
        self.nq, self.total, self.max_pairs = nq, total, max_pairs
        self.pair_off, self.cand_ids, self.hits, self.ucount, self.dev = pair_off, cand_ids, hits, ucount, dev
        self.segments = segments
        self.big = list(big)             # queries whose pair lists went through global memory (longer than the LDS network)


def _scan(torch, lib, counts, nq, dev, stream, top_k=-1, top_p=-1.0, keep_out=None):
    offsets = torch.empty(nq + 1, dtype=torch.int64, device=dev)
    totals = torch.empty(2, dtype=torch.int64, device=dev)
    _native.check(lib.lshrs_query_scan_i32(counts.data_ptr(), nq, int(top_k), float(top_p),
                                           keep_out.data_ptr() if keep_out is not None else None, offsets.data_ptr(),
                                           totals.data_ptr(), stream), "lshrs_query_scan_i32")
    return offsets, totals


def candidates_from_index(keys_dev, desc, nseg: int, max_id: int, *, want_hits: bool = False) -> _Lists:
    """Lookup + collide for ``keys_dev`` (q, bands, B) uint8 on the device against the mirrored segments."""
    torch = _native.require_gpu()
    lib = _native.load()
    dev = keys_dev.device
    nq, nb, bb = (int(v) for v in keys_dev.shape)
    if bb > 6:
        raise TooLarge("band keys wider than 6 bytes have no codes")
    if max_id >= (1 << (63 - _bbits(nb))):
        raise TooLarge("member ids do not fit the item layout")
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        nslots = nb * nseg
        slot_start = torch.empty(max(1, nq * nslots), dtype=torch.int64, device=dev)
        slot_len = torch.empty(max(1, nq * nslots), dtype=torch.int32, device=dev)
        slot_off = torch.empty(max(1, nq * nslots), dtype=torch.int32, device=dev)
        pair_count = torch.empty(nq, dtype=torch.int32, device=dev)
        kd = keys_dev if keys_dev.is_contiguous() else keys_dev.contiguous()
        _native.check(lib.lshrs_query_lookup_u8(kd.data_ptr(), nq, nb, bb, desc.data_ptr() if desc is not None else None,
                                                nseg, slot_start.data_ptr(), slot_len.data_ptr(), slot_off.data_ptr(),
                                                pair_count.data_ptr(), stream), "lshrs_query_lookup_u8")
        pair_off, totals = _scan(torch, lib, pair_count, nq, dev, stream)
        total, max_pairs = (int(v) for v in totals.cpu().tolist())         # (the one size the host must know: what to allocate)
        cap = _native.QUERY_MAX_PAIRS
        if max_pairs >= (1 << 31) - 1:
            raise TooLarge(f"a query's buckets hold {max_pairs} members or more")
        cand_ids = torch.empty(max(1, total), dtype=torch.int64, device=dev)
        hits = torch.empty(max(1, total), dtype=torch.int32, device=dev) if want_hits else None
        ucount = torch.empty(nq, dtype=torch.int32, device=dev)
        _native.check(lib.lshrs_query_collide_index_i64(desc.data_ptr() if desc is not None else None, nseg, nb,
                                                        slot_start.data_ptr(), slot_len.data_ptr(), slot_off.data_ptr(),
                                                        pair_off.data_ptr(), nq, min(max_pairs, cap), cand_ids.data_ptr(),
                                                        hits.data_ptr() if hits is not None else None, ucount.data_ptr(),
                                                        stream), "lshrs_query_collide_index_i64")
        big = []
        if max_pairs > cap:
            # queries whose buckets hold more members than a workgroup's LDS network takes (the launch above left them ucount =
            # -1): one at a time through global memory - the same sorts, K3's long-list network (rare: 16-bit keys over tens of
            # millions of stored ids)
            counts = pair_count.cpu().numpy()
            offs = pair_off.cpu().numpy()
            big = np.flatnonzero(counts > cap).tolist()
            nbytes = int(lib.lshrs_query_big_workspace_bytes(int(counts.max())))
            if nbytes < 0:
                _native.check(nbytes, "lshrs_query_big_workspace_bytes")
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            for qi in big:
                o = int(offs[qi])
                _native.check(lib.lshrs_query_collide_big_i64(desc.data_ptr(), nseg, nb, slot_start.data_ptr() + 8 * qi * nslots,
                                                              slot_off.data_ptr() + 4 * qi * nslots, int(counts[qi]), ws.data_ptr(),
                                                              cand_ids.data_ptr() + 8 * o,
                                                              hits.data_ptr() + 4 * o if hits is not None else None,
                                                              ucount.data_ptr() + 4 * qi, stream), "lshrs_query_collide_big_i64")
            torch.cuda.current_stream(dev).synchronize()            # (the workspace dies with this frame)
    return _Lists(nq, total, min(max_pairs, cap), pair_off, cand_ids, hits, ucount, dev, nseg, big)


def candidates_from_pairs(members: np.ndarray, bands: np.ndarray, pair_off: np.ndarray, num_bands: int, dev, *,
                          want_hits: bool = False) -> _Lists:
    """Collide for pairs gathered on the host (a store with ``get_bucket`` only): ``members`` / ``bands`` flat, query i's
    at ``pair_off[i] : pair_off[i + 1]``."""
    torch = _native.require_gpu()
    lib = _native.load()
    nq = int(pair_off.shape[0]) - 1
    total = int(pair_off[-1])
    lens = np.diff(pair_off)
    max_pairs = int(lens.max()) if nq else 0
    if max_pairs > _native.QUERY_MAX_PAIRS:
        raise TooLarge(f"a query's buckets hold {max_pairs} members")
    if total and (int(members.min()) < 0 or int(members.max()) >= (1 << (63 - _bbits(num_bands)))):
        raise TooLarge("member ids do not fit the item layout")
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        m_d = torch.from_numpy(np.ascontiguousarray(members, dtype=np.int64)).to(dev)
        b_d = torch.from_numpy(np.ascontiguousarray(bands, dtype=np.int32)).to(dev)
        off_d = torch.from_numpy(np.ascontiguousarray(pair_off, dtype=np.int64)).to(dev)
        cand_ids = torch.empty(max(1, total), dtype=torch.int64, device=dev)
        hits = torch.empty(max(1, total), dtype=torch.int32, device=dev) if want_hits else None
        ucount = torch.empty(nq, dtype=torch.int32, device=dev)
        _native.check(lib.lshrs_query_collide_pairs_i64(m_d.data_ptr(), b_d.data_ptr(), off_d.data_ptr(), nq, max_pairs,
                                                        int(num_bands), cand_ids.data_ptr(),
                                                        hits.data_ptr() if hits is not None else None, ucount.data_ptr(),
                                                        stream), "lshrs_query_collide_pairs_i64")
        torch.cuda.current_stream(dev).synchronize()        # (the uploaded pairs die with this frame)
    return _Lists(nq, total, max_pairs, off_d, cand_ids, hits, ucount, dev)


def rank_and_cut(lists: _Lists, top_k: Optional[int], top_p: Optional[float], *, queries_dev=None, corpus=None,
                 cand_rows=None) -> Tuple[np.ndarray, Optional[np.ndarray], np.ndarray]:
    """The ranked, cut answer of every query as arrays ``(ids, scores or None, bounds)``: query i's ids are
    ``ids[bounds[i]:bounds[i + 1]]``.  ``top_p`` None: the collision order, first ``top_k`` (lshrs/core/main.py:619-625).
    Else the candidates scored against ``corpus`` (device (m, dim) float32; row = ``cand_rows`` entry, default the id
    itself), ordered by score, cut to ``max(1, ceil(n * top_p))`` and ``top_k`` (:646-657)."""
    torch = _native.require_gpu()
    lib = _native.load()
    dev, nq = lists.dev, lists.nq
    k_arg = -1 if top_k is None else int(top_k)
    with torch.cuda.device(dev):
        cur = torch.cuda.current_stream(dev)
        stream = cur.cuda_stream
        keep = torch.empty(nq, dtype=torch.int32, device=dev)
        out_off, _ = _scan(torch, lib, lists.ucount, nq, dev, stream, top_k=k_arg,
                           top_p=-1.0 if top_p is None else float(top_p), keep_out=keep)
        scores = err = None
        if top_p is not None and lists.total:
            rows = lists.cand_ids if cand_rows is None else cand_rows
            scores = torch.empty(max(1, lists.total), dtype=torch.float32, device=dev)
            err = torch.zeros(1, dtype=torch.int32, device=dev)
            _native.check(lib.lshrs_cosine_ragged_f32(corpus.data_ptr(), int(corpus.shape[0]), int(corpus.stride(0)),
                                                      int(corpus.shape[1]), queries_dev.data_ptr(), nq, rows.data_ptr(),
                                                      lists.pair_off.data_ptr(), lists.ucount.data_ptr(), lists.total,
                                                      scores.data_ptr(), err.data_ptr(), stream), "lshrs_cosine_ragged_f32")
        bounds = out_off.cpu().numpy()                # (waits for the scan - and whatever is in front of it - only: the rerank runs on)
        kept = int(bounds[-1])
        # ids and scores side by side in ONE block: one copy back
        packed = torch.empty(12 * max(1, kept), dtype=torch.uint8, device=dev)
        out_ids = packed[:8 * kept].view(torch.int64)
        out_scores = packed[8 * kept:12 * kept].view(torch.float32) if scores is not None else None
        if kept:
            _native.check(lib.lshrs_query_rank_f32(lists.cand_ids.data_ptr(), scores.data_ptr() if scores is not None else None,
                                                   lists.pair_off.data_ptr(), lists.ucount.data_ptr(), keep.data_ptr(),
                                                   out_off.data_ptr(), nq, lists.max_pairs, out_ids.data_ptr(),
                                                   out_scores.data_ptr() if out_scores is not None else None, None, 0, stream),
                          "lshrs_query_rank_f32")
        if scores is not None and lists.big and kept:
            # a candidate list longer than the rank kernel's LDS network (it skipped those queries): K3's global network, per query
            from .similarity import topk_desc_device

            ucounts = lists.ucount.cpu().numpy()
            offs = lists.pair_off.cpu().numpy()
            keeps = np.diff(bounds)
            for qi in lists.big:
                u, k_q = int(ucounts[qi]), int(keeps[qi])
                if u <= _native.QUERY_MAX_PAIRS or k_q == 0:
                    continue
                o, ob = int(offs[qi]), int(bounds[qi])
                order, srt = topk_desc_device(scores[o:o + u].view(1, u), k_q)
                out_ids[ob:ob + k_q] = lists.cand_ids[o:o + u][order.view(-1).to(torch.int64)]
                out_scores[ob:ob + k_q] = srt.view(-1)
        if scores is not None:
            host = packed[:12 * kept].cpu().numpy()
            code = int(err.item())
            if code & 5:
                raise ValueError("Cannot normalize zero vector")
            if code & 2:
                raise IndexError("candidate index out of range of the corpus")
            return host[:8 * kept].view(np.int64), host[8 * kept:].view(np.float32), bounds
        return out_ids.cpu().numpy(), None, bounds


class OneQuery:
    """ONE query - ``LSHRS.get_top_k`` / ``get_above_p`` / ``query``, the reference's own calling pattern
    (lshrs/core/main.py:524-658) - as one chain of launches with NO size read back in between and ONE wait at its end: the
    vector goes into pinned memory, the one-launch signature kernel leaves its keys on the device, ``lshrs_query_one_u8`` (lookup, pair
    list, count, order and cut in one workgroup) [and the rerank + rank launches] follow on the same stream with fixed capacities (``CAP`` pairs: a longer list leaves ``ucount = -1`` and the
    caller counts on the host), the answer lands in pinned memory and the last kernel publishes an epoch the host polls
    (``lshrs_wait_done``).  ~50 us against ~250 for the host-counted path.  One query at a time per instance (a lock)."""

    CAP = _native.QUERY_MAX_PAIRS

    def __init__(self, hasher, dev) -> None:
        torch = _native.require_gpu()
        self.dev = dev
        self.lock = threading.Lock()
        nb, bb, dim = hasher.num_bands, hasher.band_bytes, hasher.dim
        self.shape = (nb, bb, dim)
        self.ldx = (dim + 31) // 32 * 32
        pin = lambda n, dt: torch.zeros(n, dtype=dt).pin_memory()      # noqa: E731
        self.pin_x = pin(self.ldx, torch.float32)
        self.pin_flags = pin(16, torch.uint8)
        self.pin_i64 = pin(4, torch.int64)                 # out_off: [0] 0, [1] kept, [2] candidates found (-1: beyond the capacity)
        self.pin_i32 = pin(8, torch.int32)                 # [1] err, [2] done
        self.pin_ids = pin(self.CAP, torch.int64)
        self.pin_scores = pin(self.CAP, torch.float32)
        self.h_x, self.h_flags = self.pin_x.numpy(), self.pin_flags.numpy()
        self.h_i64, self.h_i32 = self.pin_i64.numpy(), self.pin_i32.numpy()
        self.h_ids, self.h_scores = self.pin_ids.numpy(), self.pin_scores.numpy()
        with torch.cuda.device(dev):
            self.hash_counters = torch.zeros(1, dtype=torch.int32, device=dev)
            # the query's keys: from the signature launch to the lookup launch behind it, on the device (nobody on the host reads
            # them: in pinned memory they cost the lookup a round trip over the link before its first bucket)
            self.keys_dev = torch.zeros(max(16, nb * bb), dtype=torch.uint8, device=dev)
            self.pair_count = torch.zeros(1, dtype=torch.int32, device=dev)
            self.pair_off = torch.zeros(2, dtype=torch.int64, device=dev)
            self.keep = torch.zeros(1, dtype=torch.int32, device=dev)
            self.ucount = torch.zeros(1, dtype=torch.int32, device=dev)
            self.zero_off = torch.zeros(2, dtype=torch.int64, device=dev)      # (one query: its results start at 0)
            self.x_dev = torch.zeros(self.ldx, dtype=torch.float32, device=dev)      # the query vector, for the rerank launch
            self.cand_ids = torch.empty(self.CAP, dtype=torch.int64, device=dev)
            self.scores = torch.empty(self.CAP, dtype=torch.float32, device=dev)
            self.slots = None
        self.epoch = 0
        p32 = self.pin_i32.data_ptr()
        self.ptr = dict(x=self.pin_x.data_ptr(), keys=self.keys_dev.data_ptr(), flags=self.pin_flags.data_ptr(),
                        out_off=self.pin_i64.data_ptr(), ucount=self.ucount.data_ptr(), zero_off=self.zero_off.data_ptr(), err=p32 + 4,
                        done=p32 + 8, ids=self.pin_ids.data_ptr(),
                        scores_out=self.pin_scores.data_ptr(), hash_counters=self.hash_counters.data_ptr(),
                        pair_count=self.pair_count.data_ptr(), pair_off=self.pair_off.data_ptr(), keep=self.keep.data_ptr(),
                        cand=self.cand_ids.data_ptr(), scores=self.scores.data_ptr(), x_dev=self.x_dev.data_ptr())

    @staticmethod
    def applies(hasher) -> bool:
        """Where the one-launch signature kernel serves (``LSHHasher._hash_small_locked``'s conditions)."""
        return (getattr(hasher, "tie_replay", None) == "auto" and getattr(hasher, "tie_break", None) == "host"
                and hasher.dim % 4 == 0 and 8 <= hasher.dim <= 4096 and hasher.rows_per_band != 1
                and hasher._replay_model() in (1, 2))

    def run(self, hasher, vec: np.ndarray, desc, nseg: int, max_id: int, top_k: int, top_p: float, corpus):
        """(number of candidates or -1, ids int64[keep], scores float32[keep] or None, row flag).  ``top_k`` -1 = all,
        ``top_p`` -1.0 = no rerank."""
        torch = _native.require_gpu()
        lib = _native.load()
        nb, bb, dim = self.shape
        if max_id >= (1 << (63 - _bbits(nb))):
            raise TooLarge("member ids do not fit the item layout")
        dev = self.dev
        p = self.ptr
        with self.lock, torch.cuda.device(dev):
            raw = torch._C._cuda_getCurrentRawStream(dev.index)
            nslots = nb * nseg
            if self.slots is None or self.slots[0] < nslots:
                self.slots = (nslots, torch.empty(max(1, nslots), dtype=torch.int64, device=dev),
                              torch.empty(max(1, nslots), dtype=torch.int32, device=dev),
                              torch.empty(max(1, nslots), dtype=torch.int32, device=dev))
            _, s_start, s_len, s_off = self.slots
            with hasher._lock:                       # (the hasher's caches: the device image of the hyperplanes, the order model)
                ws = hasher._workspace(dev)
                model = hasher._replay_model()
            self.h_x[:dim] = vec
            self.h_i32[1] = 0
            self.epoch = epoch = self.epoch % 0x7FFFFFF0 + 1
            rc = lib.lshrs_sig_hash_small_replay_f32(p["x"], 1, self.ldx, ws.data_ptr(), nb, hasher.rows_per_band, dim, p["keys"],
                                                     p["flags"], p["hash_counters"], float(8.0 * _U), model, None, 0, raw)
            rerank = top_p >= 0.0
            rc = rc or lib.lshrs_query_one_u8(p["keys"], nb, bb, desc.data_ptr() if desc is not None else None, nseg,
                                              s_start.data_ptr(), s_len.data_ptr(), s_off.data_ptr(), self.CAP, int(top_k),
                                              float(top_p), 1 if rerank else 0, p["pair_off"], p["cand"], p["ucount"], p["keep"],
                                              p["out_off"], p["ids"], p["done"], epoch, p["x"] if rerank else None,
                                              p["x_dev"] if rerank else None, dim if rerank else 0, raw)
            if rerank and not rc:
                rc = lib.lshrs_cosine_ragged_f32(corpus.data_ptr(), int(corpus.shape[0]), int(corpus.stride(0)), dim, p["x_dev"], 1,
                                                 p["cand"], p["pair_off"], p["ucount"], 4096, p["scores"], p["err"], raw)
                rc = rc or lib.lshrs_query_rank_f32(p["cand"], p["scores"], p["pair_off"], p["ucount"], p["keep"], p["zero_off"], 1,
                                                    self.CAP, p["ids"], p["scores_out"], p["done"], epoch, raw)
            if rc:
                torch.cuda.current_stream(dev).synchronize()
                _native.check(rc, "the one-query chain")
            rc = lib.lshrs_wait_done(p["done"], epoch, 2_000_000, raw)
            if rc:
                _native.check(rc, "lshrs_wait_done")
            flag = int(self.h_flags[0])
            ucount = int(self.h_i64[2])
            kept = int(self.h_i64[1])
            err = int(self.h_i32[1])
            ids = self.h_ids[:kept].copy()
            scores = self.h_scores[:kept].copy() if rerank else None
        if rerank and ucount > 0 and not (flag & 1):
            if err & 5:
                raise ValueError("Cannot normalize zero vector")
            if err & 2:
                raise IndexError("candidate index out of range of the corpus")
        return ucount, ids, scores, flag
