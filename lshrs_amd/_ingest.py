"""Pipelined ingest behind ``LSHRS.index`` / ``create_signatures`` (round 5).

The reference hashes and enqueues vector by vector and flushes its operation buffer every ``buffer_size`` operations
(lshrs/core/main.py:442-518, :1113-1143).  Here a UNIT - one ``index()`` call's rows, or one loader batch of
``create_signatures`` - goes through three overlapped steps:

  1. host -> device copy and signature pass, chunk by chunk (``LSHHasher.hash_batch_packed(device_sink=...)``: the keys
     never come back to the host);
  2. per chunk, on the device and under the NEXT chunk's copy: the keys grouped into buckets (counting sort per band,
     ``DeviceCSRJob``), counts and members into pinned host blocks;
  3. per chunk, on the lane's own thread while the next chunk is on the link: the pinned blocks turned into a ``BucketCSR``
     of arrays and handed to the storage (``batch_add_csr``) - strictly in row order.  (A second Python thread for this
     step was measured and dropped: the two threads hand the interpreter lock back and forth, 37 ms instead of 29 for the
     copy loop of 500 000 x 768 rows; the lane thread has the time - it waits for the link two thirds of every chunk.)

With ``LSHRS(devices=[...])`` whole units are dealt round-robin to one LANE per entry (its own hasher, streams, device
buffers and finisher), and a unit's buckets are only handed to the storage once every earlier unit's have been: the
storage sees exactly the sequence a single device produces.  No collective, nothing shared but the read-only hyperplanes.

Error timing (lshrs/core/main.py:398, :1083): the rows in front of the first bad row (negative id, zero vector) are
stored, then the reference's ``ValueError`` is raised; nothing behind it is stored - units that were already being hashed
on other lanes are dropped.
"""

from __future__ import annotations

import threading
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional

import numpy as np

from . import _native, numa
from .packed_ops import DeviceCSRJob, bucket_csr

_ZERO_BIT = 1


class _Unit:
    """One unit of work and the baton that orders its commits behind the previous unit's."""

    __slots__ = ("seq", "ids", "arr", "prev_done", "done", "error")

    def __init__(self, seq: int, ids: np.ndarray, arr: np.ndarray, prev_done: threading.Event) -> None:
        self.seq, self.ids, self.arr, self.prev_done = seq, ids, arr, prev_done
        self.done = threading.Event()
        self.error: Optional[BaseException] = None


class CsrIngest:
    """``submit(ids, rows)`` any number of units, then ``drain()``; use as a context manager.  ``hashers``: one per lane."""

    chunk_rows = 131_072       # rows per chunk of the copy / pass / grouping pipeline inside a unit

    def __init__(self, hashers: List, sink, zero_error, neg_error, *, max_in_flight: Optional[int] = None,
                 inline: bool = False) -> None:
        self._hashers = list(hashers)
        self._sink = sink
        self._zero_error, self._neg_error = zero_error, neg_error
        lanes = len(self._hashers)
        # inline (one lane, one unit at a time: a single index() call): the unit runs on the caller's thread, no pool at all
        self._inline = bool(inline) and lanes == 1
        self._hash_pools = [] if self._inline else [
            ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"lshrs-ingest-{i}") for i in range(lanes)]
        self._slots = threading.Semaphore(max_in_flight if max_in_flight is not None else 2 * lanes)
        self._seq = 0
        first = threading.Event()
        first.set()
        self._last_done = first
        self._futures: list = []
        self._failed: Optional[BaseException] = None
        self._fail_lock = threading.Lock()
        self.units = 0
        self.chunks = 0
        self.lane_nodes: dict = {}            # lane -> NUMA node its worker thread was bound to (None: unbound)

    # ------------------------------------------------------------------ public
    def __enter__(self) -> "CsrIngest":
        return self

    def __exit__(self, exc_type, exc, tb) -> None:
        self.close()

    def close(self, wait: bool = True) -> None:
        """Wait for everything submitted and raise the first failure in row order.  (An exception of the CALLER's - its loader
        raised - does not undo what it had already handed over: those units are stored, as the reference's sequential loop would
        have stored them.)  ``wait=False`` - the caller is being interrupted: units not yet started are dropped, the ones in
        flight are given ``interrupt_grace`` seconds, nothing is raised from here."""
        try:
            if wait:
                self._wait_all()
                if self._failed is not None:
                    raise self._failed
            else:
                self._fail(KeyboardInterrupt())            # (lanes stop at their next chunk; queued units return at once)
                for f in self._futures:
                    f.cancel()
                self._last_done.wait(self.interrupt_grace)
        finally:
            for pool in self._hash_pools:
                pool.shutdown(wait=wait, cancel_futures=not wait)

    interrupt_grace = 5.0

    @property
    def failed(self) -> bool:
        return self._failed is not None

    def submit(self, ids: np.ndarray, arr: np.ndarray) -> None:
        """Queue one unit (``ids`` int64, ``arr`` (n, dim) float32, same length).  Blocks while ``max_in_flight`` units are
        unfinished.  After a failure nothing more is accepted (the caller stops reading its loader and drains)."""
        if self._failed is not None or len(ids) == 0:
            return
        self._slots.acquire()
        unit = _Unit(self._seq, ids, arr, self._last_done)
        self._last_done = unit.done
        lane = self._seq % len(self._hashers)
        self._seq += 1
        self.units += 1
        if self._inline:
            self._run_unit(lane, unit)
        else:
            self._futures.append(self._hash_pools[lane].submit(self._run_unit, lane, unit))

    def drain(self) -> None:
        """Wait for everything submitted; raise the first error (in row order) if there was one."""
        self._wait_all()
        if self._failed is not None:
            raise self._failed

    # ------------------------------------------------------------------ internals
    def _wait_all(self) -> None:
        for f in self._futures:
            try:
                f.result()
            except BaseException as exc:  # noqa: BLE001 - recorded; the first in row order is what drain() raises
                self._fail(exc)
        self._futures = []
        self._last_done.wait()

    def _fail(self, exc: BaseException) -> None:
        with self._fail_lock:
            if self._failed is None:
                self._failed = exc

    def _run_unit(self, lane: int, unit: _Unit) -> None:
        """Lane thread: stream the unit through the hasher; every chunk's grouping is enqueued on the device by the sink
        callback, which first finishes and commits the chunk BEFORE it (whose device work ended a chunk's copy ago)."""
        torch = _native.require_gpu()
        hasher = self._hashers[lane]
        if not self._inline:        # a lane's own worker thread: onto the CPUs of its GPU's NUMA node, with the pinned blocks it
            dev = hasher._torch_device()       # allocates from here on (lshrs_amd/numa.py; never the caller's thread)
            self.lane_nodes[lane] = numa.bind_current_thread(dev.index if dev.index is not None else 0)
        ids, arr = unit.ids, unit.arr
        n = int(ids.shape[0])
        negs = np.flatnonzero(ids < 0)
        limit = int(negs[0]) if negs.size else n          # rows from the first negative id on are never stored
        state = {"stop": limit, "error": self._neg_error() if limit < n else None, "committed_to": 0}
        pending: list = []
        resident = isinstance(arr, torch.Tensor) and arr.is_cuda      # (vectors that live on a GPU: hashed where they are)

        def commit(job, lo, hi):
            """Chunk order: bucket arrays of rows lo:hi -> storage, unless a bad row lies at or before them."""
            try:
                csr = job.finish()
                flags = job.flags_host()
                bad = np.flatnonzero(flags & _ZERO_BIT) if flags is not None else np.empty(0, np.int64)
                if state["stop"] < hi:                     # an earlier chunk (or the negative id) ends the unit in front of hi
                    return
                if bad.size:
                    state["stop"], state["error"] = lo + int(bad[0]), self._zero_error()
                    return
                unit.prev_done.wait()
                if self._failed is None:
                    self._sink.batch_add_csr(csr)
                    state["committed_to"] = hi
            except BaseException as exc:  # noqa: BLE001
                state["stop"], state["error"] = min(state["stop"], lo), exc

        def sink(lo, hi, keys_dev, flags_dev=None):
            # Enqueue the chunk's grouping, wait for it (~1 ms of device work, in front of nothing: the stream is the one the
            # keys would have travelled back on) and finish it right here: the NEXT chunk's copy is already on the link and takes
            # longer than all of this, so the link never waits - and behind the last chunk only ITS finish is left.
            job = _ChunkJob(ids[lo:hi], keys_dev, flags_dev, ids_dev[lo:hi])
            self.chunks += 1
            if resident:
                # vectors that live on the GPU: there is no copy for this chunk's grouping to hide under - the chunk BEFORE it is
                # finished and stored (host work) while the device groups this one
                pending.append((job, lo, hi))
                if len(pending) > 1:
                    commit(*pending.pop(0))
            else:
                commit(job, lo, hi)
            # (the hasher calls this WITHOUT its lock - queries go on while a chunk is finished, waits for its turn and is
            #  stored.)  A unit that is known to end inside or in front of this chunk - a zero vector, a storage failure, another
            #  lane's failure - needs nothing behind it: False ends the stream instead of copying and hashing rows nobody stores
            return not (state["stop"] < hi or self._failed is not None)

        try:
            if limit > 0 and self._failed is None:
                dev = arr.device if resident else hasher._torch_device()
                with torch.cuda.device(dev):
                    ids_dev = torch.from_numpy(np.ascontiguousarray(ids[:limit])).to(dev)   # once, in front of the stream
                    hasher.hash_batch_packed(arr[:limit], return_row_flags=True, device_sink=sink, chunk_rows=self.chunk_rows)
            while pending:
                commit(*pending.pop(0))
            # a unit that ends inside a chunk (zero vector at row `stop`): the rows of that chunk in front of it, once more,
            # synchronously - rare, and the only place the keys of a chunk are needed twice
            stop, done_to = state["stop"], state["committed_to"]
            if state["error"] is not None and not isinstance(state["error"], ValueError):
                raise state["error"]
            if stop > done_to:
                unit.prev_done.wait()
                if self._failed is None:
                    tail_dev = arr.device if resident else hasher._torch_device()
                    with torch.cuda.device(tail_dev):
                        keys = hasher.hash_batch_packed(arr[done_to:stop])
                        self._sink.batch_add_csr(bucket_csr(ids[done_to:stop], keys, device=tail_dev))
            if state["error"] is not None:
                unit.prev_done.wait()
                self._fail(state["error"])
        except BaseException as exc:  # noqa: BLE001
            unit.prev_done.wait()
            self._fail(exc)
        finally:
            unit.ids = unit.arr = None
            unit.prev_done.wait()          # (never overtake: a unit is done only when every earlier one is)
            unit.done.set()
            self._slots.release()


class _ChunkJob(DeviceCSRJob):
    """A chunk's grouping plus its row flags (bit0 = zero vector) on their way to the host."""

    def __init__(self, ids, keys_dev, flags_dev, ids_dev=None) -> None:
        torch = _native.require_gpu()
        self._flags_h = None
        self._csr = None
        if flags_dev is not None:       # (in front of the job's event, on the same stream; by a kernel like the job's own results)
            self._flags_h = torch.empty(flags_dev.shape[0], dtype=torch.uint8, pin_memory=True)
            fd = flags_dev.contiguous()
            _native.check(_native.load().lshrs_copy_to_host_u8(fd.data_ptr(), self._flags_h.data_ptr(), fd.numel(),
                                                               torch.cuda.current_stream(fd.device).cuda_stream),
                          "lshrs_copy_to_host_u8")
        if int(keys_dev.shape[2]) > 2:
            # keys of 3 bytes and more (config 5: 4) do not fit the counting sort's table: the device sort of `bucket_csr`,
            # which waits for its own results - the chunk's grouping is then not under the next chunk's copy
            self._csr = bucket_csr(ids, keys_dev)
            self.event = torch.cuda.Event()
            self.event.record(torch.cuda.current_stream(keys_dev.device))
        else:
            super().__init__(ids, keys_dev, ids_dev)

    def finish(self):
        if self._csr is None:
            return super().finish()
        self.event.synchronize()
        return self._csr

    def flags_host(self):
        return None if self._flags_h is None else self._flags_h.numpy()
