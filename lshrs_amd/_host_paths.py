"""The host-facing half of ``LSHHasher`` (mixin): NumPy arrays in, NumPy keys / ``HashSignatures`` out - the one-launch path
for a vector or a handful, the three-stream path for large arrays, in-process multi-device slicing, the coalescing of
concurrent single-vector callers, and the reference's own entry points ``hash_vector`` / ``hash_batch``
(lshrs/hash/lsh.py:96-169)."""

from __future__ import annotations

import contextlib
import threading  # noqa: F401
from typing import List, Optional

import numpy as np

from . import _native, numa
from ._config import HashSignatures
from .windows import _U


class _OneRequest:
    """A single-vector call waiting to be hashed (see :meth:`_HostPaths.hash_one_packed`)."""

    __slots__ = ("vec", "keys", "flag", "error", "event", "lead")

    def __init__(self, vec) -> None:
        self.vec, self.keys, self.flag, self.error, self.lead = vec, None, 0, None, False
        self.event = threading.Event()


class _HostPaths:
    # ------------------------------------------------------------------ host-facing API
    def hash_batch_packed(self, vectors, *, return_row_flags: bool = False, chunk_rows: int = 131_072,
                          tie_break: Optional[str] = None, pin: str = "auto", device_sink=None):
        """Hash host vectors; returns a NumPy ``(n, num_bands, band_bytes)`` uint8 array
        (the reference's ``bytes`` keys side by side) and, on request, the per-row flag byte.

        Host-resident input is bound by the PCIe link (63 GB/s spec = 20.5 M vec/s at 768-d), not by the kernel, so
        large batches are streamed: chunk i+1 crosses the link on a copy stream while chunk i is hashed and the keys
        of chunk i-1 travel back on a third stream, through two device buffers and two pinned key buffers per hasher.
        ``pin``: "auto" page-locks a large pageable source array in place for the duration of the call
        (``hipHostRegister``; the DMA engines then read it directly instead of going through the runtime's staging
        copies), "never" leaves it to the runtime; a source that is already pinned is used as it is.
        ``device_sink(lo, hi, keys_dev, flags_dev)``: the keys do NOT come back to the host; instead the callable is handed
        every finished chunk's keys (and row flags, or None) ON THE DEVICE (rows ``lo:hi`` of the batch, final and verified;
        the current stream is one it may enqueue work on, and the tensors stay valid for work enqueued there) - ``LSHRS.index``
        groups them into buckets on the device under the next chunk's copy.  The return value is then ``(None, row_flags)``
        / ``None``.  The sink runs WITHOUT the hasher's lock (round 6: it may wait for other work, write to a remote store, call
        back into the hasher - queries go on meanwhile; what it holds is the streaming lock: one streamed batch per hasher at a
        time, their staging buffers are shared); a sink that returns ``False`` ends the stream - the chunks in flight are
        finished and discarded, nothing more is copied or hashed."""
        torch = _native.require_gpu()
        if isinstance(vectors, torch.Tensor) and vectors.is_cuda:      # vectors that already live on a GPU: nothing crosses the link
            return self._hash_resident(vectors, return_row_flags, chunk_rows, device_sink)
        arr = np.asarray(vectors, dtype=np.float32)
        if arr.ndim != 2:
            raise ValueError("Batch input must be a 2D array")
        if arr.shape[1] != self.dim:
            raise ValueError(f"Expected vectors of dimension {self.dim}, received {arr.shape[1]}")
        arr = np.ascontiguousarray(arr)
        if not arr.flags.writeable:  # torch.from_numpy refuses read-only buffers (np.frombuffer views)
            arr = arr.copy()
        n = arr.shape[0]
        mode = self.tie_break if tie_break is None else tie_break
        if (self._devices is not None and len(self._devices) > 1 and tie_break is None and device_sink is None
                and n >= self.multi_device_min_rows * len(self._devices)):
            return self._hash_multi_device(arr, return_row_flags, chunk_rows, pin)
        dev = self._torch_device()
        with self._lock:
            streamed = (n >= 2 * 16_384 and mode == "host" and self.tie_replay == "auto"
                        and self._split_applies(16_384, replay=True) and self._replay_model() in (1, 2))
        if streamed:
            # one streamed batch per hasher at a time (the staging buffers are shared); the hasher's own lock is only taken
            # around each chunk's launch and verification - not around copies, waits or the sink
            with self._stream_lock:
                return self._hash_host_streamed(arr, return_row_flags, max(16_384, int(chunk_rows)), dev, pin, device_sink)
        if 0 < n <= self._small_rows and mode == "host" and device_sink is None:
            with self._lock:
                got = self._hash_small_locked(arr, dev)
            if got is not None:
                return got if return_row_flags else got[0]
        keys = np.empty((n, self.num_bands, self.band_bytes), dtype=np.uint8) if device_sink is None else None
        flags = np.empty(n, dtype=np.uint8) if return_row_flags else None
        total = {"n": n, "tie_entries": 0, "tie_pairs": 0, "relaunches": 0}
        for lo in range(0, n, 262_144):
            hi = min(n, lo + 262_144)
            chunk = arr[lo:hi]
            x = torch.from_numpy(chunk).to(dev)
            fl = torch.empty(hi - lo, dtype=torch.uint8, device=dev) if return_row_flags else None
            with self._lock:
                out = self._hash_device_locked(x, None, fl, mode, host_rows=lambda r, c=chunk: c[r])
                for k in ("tie_entries", "tie_pairs", "relaunches"):
                    total[k] += self.last_stats.get(k, 0)
            if device_sink is not None:                  # (outside the lock: the keys are a tensor of this call's own)
                with torch.cuda.device(dev):
                    if device_sink(lo, hi, out, fl) is False:
                        break
            else:
                keys[lo:hi] = out.cpu().numpy()
            if fl is not None:
                flags[lo:hi] = fl.cpu().numpy()
        with self._lock:
            self.last_stats = total
        return (keys, flags) if return_row_flags else keys

    def _hash_resident(self, x, return_row_flags: bool, chunk_rows: int, device_sink):
        """:meth:`hash_batch_packed` for a torch tensor on a GPU (round 6): `hash_device` chunk by chunk on the tensor's own
        device - the keys to the sink as they are (``LSHRS.index`` of device-resident vectors: the buckets are grouped on the
        device, only the bucket arrays travel), or back to the host as the array the other form returns."""
        torch = _native.require_gpu()
        if x.dim() != 2:
            raise ValueError("Batch input must be a 2D array")
        if int(x.shape[1]) != self.dim:
            raise ValueError(f"Expected vectors of dimension {self.dim}, received {int(x.shape[1])}")
        if x.dtype != torch.float32:
            x = x.float()
        if x.stride(1) != 1 or x.stride(0) < x.shape[1]:
            x = x.contiguous()
        n, dev = int(x.shape[0]), x.device
        keys = np.empty((n, self.num_bands, self.band_bytes), dtype=np.uint8) if device_sink is None else None
        flags = np.empty(n, dtype=np.uint8) if return_row_flags and device_sink is None else None
        step = 4 * max(16_384, int(chunk_rows))        # (no copy to pace: larger chunks than the host-fed stream's - fewer waits)
        with torch.cuda.device(dev):
            for lo in range(0, n, step):
                hi = min(n, lo + step)
                fl = torch.empty(hi - lo, dtype=torch.uint8, device=dev) if (return_row_flags or device_sink is not None) else None
                out = self.hash_device(x[lo:hi], row_flags=fl)           # (final and verified; takes the hasher's lock itself)
                if device_sink is not None:
                    if device_sink(lo, hi, out, fl) is False:
                        break
                else:
                    keys[lo:hi] = out.cpu().numpy()
                    if flags is not None:
                        flags[lo:hi] = fl.cpu().numpy()
        return (keys, flags) if return_row_flags else keys

    def device_hashers(self) -> list:
        """One hasher per entry of ``devices`` (own device, streams, scratch; this hasher's hyperplanes) - the lanes of the
        pipelined ingest (``lshrs_amd/_ingest.py``) and of ``_hash_multi_device``; ``[self]`` for a single-device hasher."""
        if self._devices is None or len(self._devices) < 2:
            return [self]
        with self._lock:
            return list(self._device_hashers_locked())

    def _device_hashers_locked(self) -> list:
        devs = self._devices
        if self._children is None:
            self._children = [type(self)(self.num_bands, self.rows_per_band, self.dim, self._seed, device=d,
                                        **self._ctor_kwargs) for d in devs]
            self._child_version = [-1] * len(devs)
        for i, c in enumerate(self._children):        # the hyperplanes are the parent's (assignable: load_from_disk)
            if self._child_version[i] != self._projection_version:
                c.projections = [np.asarray(p) for p in self._projections]
                self._child_version[i] = self._projection_version
        return self._children

    def _hash_multi_device(self, arr: np.ndarray, want_flags: bool, chunk_rows: int, pin: str):
        """One contiguous row slice per entry of ``devices``, each through its own hasher on its own thread; keys (and
        row flags) of all slices in one array, in the original row order (the caller enqueues storage operations from
        it exactly as from a single-device result: lshrs/core/main.py:442-518 sees no difference)."""
        from concurrent.futures import ThreadPoolExecutor

        torch = _native.require_gpu()
        devs = self._devices
        with self._lock:
            self._device_hashers_locked()
            if self._pool is None:
                self._pool = ThreadPoolExecutor(max_workers=len(devs), thread_name_prefix="lshrs-dev")
            n = arr.shape[0]
            keys = np.empty((n, self.num_bands, self.band_bytes), dtype=np.uint8)
            flags = np.empty(n, dtype=np.uint8) if want_flags else None
            per = -(-n // len(devs))
            per = -(-per // 256) * 256                        # whole 256-row workgroup tiles per slice
            spans = [(lo, min(n, lo + per)) for lo in range(0, n, per)]

            def work(i):
                lo, hi = spans[i]
                child = self._children[i]
                numa.bind_current_thread(devs[i])        # (a worker thread of the pool: onto its device's NUMA node)
                with torch.cuda.device(devs[i]):
                    got = child.hash_batch_packed(arr[lo:hi], return_row_flags=want_flags, chunk_rows=chunk_rows, pin=pin)
                if want_flags:
                    keys[lo:hi], flags[lo:hi] = got
                else:
                    keys[lo:hi] = got
                return dict(child.last_stats)

            stats = [f.result() for f in [self._pool.submit(work, i) for i in range(len(spans))]]
            total = {"n": n, "devices": [devs[i] for i in range(len(spans))], "per_device": stats}
            for k in ("tie_entries", "tie_pairs", "relaunches", "flagged", "margin_escalations"):
                total[k] = sum(int(st.get(k, 0)) for st in stats)
            total["max_dev_units"] = max(float(st.get("max_dev_units", 0.0)) for st in stats)
            total["tie_break_engine"] = stats[0].get("tie_break_engine")
            self.last_stats = total
        return (keys, flags) if want_flags else keys

    _small_rows = 128                 # batches up to this many host rows take the one-launch path below
    _small_poll_bytes = 128           # ... up to this many KEY BYTES (= workgroups) return through the polled epoch word
    _small_direct_bytes = 2 << 20     # ... and x is read by the kernel straight from pinned host memory while the key
                                      #     bytes' reads of it (one per key byte and row) stay below this many bytes
                                      #     (both thresholds: profiles/r02_small_latency.log)

    def _hash_small_locked(self, arr: np.ndarray, dev):
        """A query vector or a handful (``ingest``, ``get_top_k``, ``hash_vector``: the reference's one-vector-per-call
        pattern, lshrs/core/main.py:405,1101): ONE kernel launch.  Every projection is evaluated the way the host BLAS
        evaluates it (``lshrs_sig_hash_small_replay_f32``: the replay that decides the flagged projections of a large
        batch, applied to all of them), so the keys are the reference's with no first pass and no tie list.  The rows
        are copied into pinned host memory; a few are read by the kernel from there (3 KB per key byte over PCIe), more
        go to the device in one asynchronous copy first.  Up to ``_small_poll_bytes`` key bytes the kernel stores keys and row
        flags straight into pinned host memory and its last workgroup publishes this call's epoch there - the host
        polls that word: no copy back, no stream wait; above, one asynchronous copy back and one wait (the per-workgroup
        system-scope fences of the polled form cost more than they save there).  Returns None where the replay does not
        apply (host tie-break engine, ``dim % 4``, ``dim > 4096``): the general path then does the work."""
        torch = _native.require_gpu()
        lib = _native.load()
        n = arr.shape[0]
        rb = self.num_bands * self.band_bytes
        if self.tie_replay != "auto" or self.dim % 4 != 0 or self.dim < 8 or self.dim > 4096 or self.rows_per_band == 1:
            return None
        ldx = (self.dim + 31) // 32 * 32          # (the kernel fetches whole k-tiles: rows padded with zeros, never used)
        model = self._replay_model()
        if not model or model == 3:      # (model 3: the SkylakeX build's small-matrix kernels - the general path's plain-load replay)
            return None
        key = ("small", dev.index)
        buf = self._pinned_cache.get(key)
        cap = self._small_rows
        if buf is None:
            tail = (cap * rb + cap + 15) // 16 * 16              # keys | flags | pad, then 4 int32: ties (device form), done[2]
            pin_out = torch.zeros(tail + 16, dtype=torch.uint8).pin_memory()
            pin_x = torch.zeros((cap, ldx), dtype=torch.float32).pin_memory()
            with torch.cuda.device(dev):
                x_dev = torch.zeros((cap, ldx), dtype=torch.float32, device=dev)
                dev_out = torch.zeros(tail + 16, dtype=torch.uint8, device=dev)
                counters = torch.zeros(64 * (1 + _native.SMALL_MAX_ROWS), dtype=torch.int32, device=dev)
                torch.cuda.current_stream(dev).synchronize()
            host_out = pin_out.numpy()
            buf = {"pin_out": pin_out, "host_out": host_out, "words": host_out[tail:tail + 16].view(np.int32),
                   "pin_x": pin_x, "host_x": pin_x.numpy(), "x_dev": x_dev, "dev_out": dev_out, "counters": counters,
                   "tail": tail, "ties_seen": 0}
            self._pinned_cache[key] = buf
        pin_out, host_out, words, pin_x, x_dev = buf["pin_out"], buf["host_out"], buf["words"], buf["pin_x"], buf["x_dev"]
        tail = buf["tail"]
        ws = self._workspace(dev)
        buf["host_x"][:n, :self.dim] = arr
        poll = n * rb <= self._small_poll_bytes
        ctx = contextlib.nullcontext() if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)
        with ctx:
            cur = torch.cuda.current_stream(dev)
            if n * rb * self.dim * 4 <= self._small_direct_bytes:
                x_ptr = pin_x.data_ptr()
            else:
                x_dev[:n].copy_(pin_x[:n], non_blocking=True)
                x_ptr = x_dev.data_ptr()
            tau = float(8.0 * _U)        # (this kernel decides every projection itself; tau only feeds its tie statistic)
            if poll:
                epoch = self._small_epoch = self._small_epoch % 0x7FFFFFF0 + 1
                base = pin_out.data_ptr()
                _native.check(
                    lib.lshrs_sig_hash_small_replay_f32(x_ptr, n, ldx, ws.data_ptr(), self.num_bands,
                                                        self.rows_per_band, self.dim, base, base + cap * rb,
                                                        buf["counters"].data_ptr(), tau, model, base + tail + 4, epoch,
                                                        cur.cuda_stream), "lshrs_sig_hash_small_replay_f32")
                spins = 0
                while words[2] != epoch:                         # (~20 us; the stream wait is the fallback, not the path)
                    spins += 1
                    if spins > 20_000:
                        cur.synchronize()
                        if words[2] != epoch:
                            raise _native.NativeLibraryError("lshrs_sig_hash_small_replay_f32 finished without "
                                                             "publishing its epoch")
                ties = int(words[1])
            else:
                dev_out = buf["dev_out"]
                base = dev_out.data_ptr()                        # (counter [0] lives in the block that comes back: it only grows)
                _native.check(
                    lib.lshrs_sig_hash_small_replay_f32(x_ptr, n, ldx, ws.data_ptr(), self.num_bands,
                                                        self.rows_per_band, self.dim, base, base + cap * rb, base + tail,
                                                        tau, model, None, 0, cur.cuda_stream),
                    "lshrs_sig_hash_small_replay_f32")
                pin_out.copy_(dev_out, non_blocking=True)
                cur.synchronize()
                seen = int(words[0])
                ties = (seen - buf["ties_seen"]) & 0x7FFFFFFF
                buf["ties_seen"] = seen
        keys = host_out[:n * rb].reshape(n, self.num_bands, self.band_bytes).copy()
        flags = host_out[cap * rb:cap * rb + n].copy()
        self.last_stats = {"n": n, "tie_entries": ties, "tie_pairs": ties, "relaunches": 0,
                           "tie_break_engine": "device-replay", "path": "small-replay"}
        return keys, flags

    def _stream_buffers(self, dev, rows: int):
        """Two device input buffers, two device key / flag buffers and their pinned host mirrors, three streams."""
        torch = _native.require_gpu()
        key = ("stream", dev.index, rows)
        buf = self._pinned_cache.get(key)
        if buf is None:
            for old in [k for k in self._pinned_cache if k[0] == "stream" and k[1] == dev.index]:
                del self._pinned_cache[old]
            nb, bb = self.num_bands, self.band_bytes
            buf = {
                "x": [torch.empty((rows, self.dim), dtype=torch.float32, device=dev) for _ in range(2)],
                "k": [torch.empty((rows, nb, bb), dtype=torch.uint8, device=dev) for _ in range(2)],
                "f": [torch.empty((rows,), dtype=torch.uint8, device=dev) for _ in range(2)],
                "kh": [torch.empty((rows, nb, bb), dtype=torch.uint8).pin_memory() for _ in range(2)],
                "fh": [torch.empty((rows,), dtype=torch.uint8).pin_memory() for _ in range(2)],
                "copy": torch.cuda.Stream(device=dev), "compute": torch.cuda.Stream(device=dev),
                "back": torch.cuda.Stream(device=dev),
            }
            self._pinned_cache[key] = buf
        return buf

    def _hash_host_streamed(self, arr: np.ndarray, want_flags: bool, chunk_rows: int, dev, pin: str, device_sink=None):
        torch = _native.require_gpu()
        n = arr.shape[0]
        keys = np.empty((n, self.num_bands, self.band_bytes), dtype=np.uint8) if device_sink is None else None
        flags = np.empty(n, dtype=np.uint8) if want_flags and device_sink is None else None      # (a sink gets the flags too)
        total = {"n": n, "tie_entries": 0, "tie_pairs": 0, "relaunches": 0, "flagged": 0,
                 "max_dev_units": 0.0}
        src = torch.from_numpy(arr)
        registered = False
        if pin == "auto" and not src.is_pinned() and arr.nbytes >= (64 << 20):
            # page-lock the caller's array where it lies (undone below): the copy engines read it directly
            registered = int(torch.cuda.cudart().cudaHostRegister(arr.ctypes.data, arr.nbytes, 0)) == 0
        total["source"] = "pinned" if src.is_pinned() else ("registered" if registered else "pageable")
        buf = self._stream_buffers(dev, chunk_rows)
        copy_s, comp_s, back_s = buf["copy"], buf["compute"], buf["back"]
        caller = torch.cuda.current_stream(dev)
        for st in (copy_s, comp_s, back_s):
            st.wait_stream(caller)
        spans = [(lo, min(n, lo + chunk_rows)) for lo in range(0, n, chunk_rows)]
        x_free = [None, None]      # event: the pass that read x buffer b has finished
        k_free = [None, None]      # event: the keys of buffer b have reached the host
        hashed, landed = [], []    # (handle, span, slot) in flight on the GPU / on their way back

        def land(item):
            ev, (lo, hi), b = item
            ev.synchronize()
            if keys is not None:
                keys[lo:hi] = buf["kh"][b][:hi - lo].numpy()
            if flags is not None:
                flags[lo:hi] = buf["fh"][b][:hi - lo].numpy()

        aborted = [False]          # the sink has said it wants no more (a bad row ends the unit: lshrs_amd/_ingest.py)

        def send_back(item):
            handle, (lo, hi), b = item
            with self._lock, torch.cuda.stream(comp_s):      # verify (repeats the chunk with room / with a wider window if it must)
                st = handle._finish_locked() or {}
            while len(landed) > 1:                   # (the pinned key buffer b is about to be overwritten: empty it first)
                land(landed.pop(0))
            for k in ("tie_entries", "tie_pairs", "relaunches", "flagged"):
                total[k] += st.get(k, 0)
            total["max_dev_units"] = max(total["max_dev_units"], st.get("max_dev_units", 0.0))
            x_free[b] = torch.cuda.Event()
            x_free[b].record(comp_s)
            if device_sink is None:
                back_s.wait_stream(comp_s)
            # (with a sink nothing is waited for: `_finish_locked` has just seen this chunk's pass complete on the host, its keys
            #  are final - and the compute stream already holds the NEXT chunk's pass, which waits for that chunk's copy: a sink
            #  that finishes its work at once would wait a whole chunk's copy for keys that are there)
            with torch.cuda.stream(back_s):
                if device_sink is not None:       # the keys (and flags) stay on the device: the caller's work on them rides on this stream
                    if not aborted[0] and device_sink(lo, hi, buf["k"][b][:hi - lo], buf["f"][b][:hi - lo] if want_flags else None) is False:
                        aborted[0] = True
                else:
                    buf["kh"][b][:hi - lo].copy_(buf["k"][b][:hi - lo], non_blocking=True)
                if flags is not None:
                    buf["fh"][b][:hi - lo].copy_(buf["f"][b][:hi - lo], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(back_s)
            k_free[b] = ev
            landed.append((ev, (lo, hi), b))

        try:
            for i, (lo, hi) in enumerate(spans):
                b = i & 1
                if x_free[b] is not None:
                    copy_s.wait_event(x_free[b])
                with torch.cuda.stream(copy_s):
                    buf["x"][b][:hi - lo].copy_(src[lo:hi], non_blocking=True)
                    h2d = torch.cuda.Event()
                    h2d.record(copy_s)
                comp_s.wait_event(h2d)
                if k_free[b] is not None:
                    comp_s.wait_event(k_free[b])       # (the keys this pass overwrites have left the device)
                with self._lock, torch.cuda.stream(comp_s):
                    handle = self._hash_device_async_locked(buf["x"][b][:hi - lo], buf["k"][b][:hi - lo],
                                                            buf["f"][b][:hi - lo] if want_flags else None)
                hashed.append((handle, (lo, hi), b))
                if len(hashed) > 1:
                    send_back(hashed.pop(0))
                if aborted[0]:                 # (what is in flight is still verified below - its counter blocks go back - and dropped)
                    break
            while hashed:
                send_back(hashed.pop(0))
            while landed:
                land(landed.pop(0))
            caller.wait_stream(comp_s)
        finally:
            if registered:
                torch.cuda.synchronize(dev)
                torch.cuda.cudart().cudaHostUnregister(arr.ctypes.data)
        total["tie_break_engine"] = "device-replay"
        total["margin_escalations"] = self.margin_escalations
        if aborted[0]:
            total["aborted_by_sink"] = True
        with self._lock:
            self.last_stats = total
        return (keys, flags) if want_flags else keys

    def hash_one_packed(self, vec: np.ndarray):
        """One validated ``(dim,)`` float32 vector -> ``(keys (num_bands, band_bytes) uint8, row flag)``.

        A single vector costs the GPU path a host->device copy, a launch and a copy back (~85 us; the reference's 16
        small ``sgemv`` calls take ~36 us on the same host, 64 vectors 2.3 ms against 0.1 ms here - ``bench.py``
        ``small_n``), and callers of ``ingest`` / ``get_top_k`` may come from many threads at once
        (tests/test_concurrency.py of the reference).  Concurrent single-vector calls are therefore COALESCED: the first
        caller to arrive becomes the leader and hashes everything that is queued - its own vector and those of the
        threads that arrived while the previous launch was in flight - in one launch; when it is done and more have
        queued, it hands the lead to the first of them.  One launch per wave of callers instead of one each."""
        req = _OneRequest(vec)
        with self._one_lock:
            self._one_queue.append(req)
            lead = not self._one_leader
            if lead:
                self._one_leader = True
        if not lead:
            req.event.wait()
            if not req.lead:
                if req.error is not None:
                    raise req.error
                return req.keys, req.flag
        with self._one_lock:
            batch, self._one_queue = self._one_queue, []
        try:
            rows = batch[0].vec.reshape(1, -1) if len(batch) == 1 else np.stack([r.vec for r in batch])
            keys, flags = self.hash_batch_packed(rows, return_row_flags=True)
            for i, r in enumerate(batch):
                r.keys, r.flag = keys[i], int(flags[i])
        except BaseException as exc:  # noqa: BLE001 - every caller of the batch sees what went wrong
            for r in batch:
                r.error = exc
        for r in batch:
            if r is not req:
                r.event.set()
        with self._one_lock:
            if self._one_queue:
                nxt = self._one_queue[0]
                nxt.lead = True
                nxt.event.set()
            else:
                self._one_leader = False
        if req.error is not None:
            raise req.error
        return req.keys, req.flag

    def hash_vector(self, vector) -> HashSignatures:
        """One vector -> ``HashSignatures`` (reference: lsh.py:96-134)."""
        vec = self._validate_vector(vector)
        keys, _ = self.hash_one_packed(vec)
        # one `bytes` of the row, cut band by band; nothing for `__post_init__` to coerce (3 us of a 30 us call as 2 x 17 generator steps)
        raw, bb = keys.tobytes(), self.band_bytes
        sig = object.__new__(HashSignatures)
        object.__setattr__(sig, "bands", tuple([raw[i:i + bb] for i in range(0, len(raw), bb)]) if bb else tuple(b"" for _ in range(self.num_bands)))
        return sig

    def hash_batch(self, vectors) -> List[HashSignatures]:
        """``(n, dim)`` -> list of ``HashSignatures`` (reference: lsh.py:136-169)."""
        return HashSignatures._from_packed(self.hash_batch_packed(vectors))
