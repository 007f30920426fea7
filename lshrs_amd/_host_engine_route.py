"""The host-engine route of ``LSHHasher`` (mixin): what runs where the device cannot replay the host BLAS - its summation order
is not one `_hostblas.blas_order_model` recognises, or ``tie_replay="off"`` - and for ``tie_break="none"``: one signature pass
(split or f32 kernel) that REPORTS the projections inside the tie window, the tied (row, band) pairs decided by the library's
own ``cblas_sgemv`` (the host engine, or NumPy), large batches chunked and overlapped by ``csrc/pipeline.hip``."""

from __future__ import annotations

import contextlib
import ctypes
import os
import time
from typing import Tuple

import numpy as np

from . import _hostblas, _native


class _HostEngineRoute:
    # ------------------------------------------------------------------ large batches: overlap the tie-break
    def _hash_device_pipelined(self, x, out, row_flags, ws, tau, stats):
        """Large device batches whose ties the HOST breaks (BLAS order not recognised, ``tie_replay="off"``): chunked by
        the library's own driver (csrc/pipeline.hip) with the host engine's work overlapped."""
        try:
            return self._pipelined_native(x, out, row_flags, ws, tau, stats, self._tie_engine())
        except BaseException:
            # kernels and copies still in flight use buffers owned by the frame that just unwound: let them finish
            # before the caching allocator can hand that memory to anyone else
            _native.require_gpu().cuda.synchronize(x.device)
            raise

    def _redo_overflowed(self, x, out, row_flags, overflow, stats):
        for lo, hi in overflow:  # a chunk with more ties than its list holds: redo it on the plain path
            sub = {"n": hi - lo, "tie_entries": 0, "tie_pairs": 0, "relaunches": 0}
            self.last_stats = sub
            self._hash_device_locked(x[lo:hi], out[lo:hi], row_flags[lo:hi] if row_flags is not None else None,
                                     "host", None, allow_pipeline=False)
            for k in ("tie_entries", "tie_pairs"):
                stats[k] += self.last_stats[k]
            stats["relaunches"] += 1 + self.last_stats["relaunches"]
        self.last_stats = stats
        return out

    def _pipeline_plan(self, n: int):
        """(chunk rows, tie entries of room per chunk, [(lo, hi) ...]) for a device batch of n rows."""
        ch = self.pipeline_chunk_rows
        while n < 2 * ch and ch > 65_536:   # mid-size batch: two or three smaller chunks still overlap most of the tie-break
            ch = max(65_536, ch // 2)
        cap = ch // 32 + 1024    # tie entries per chunk (measured: ~0.25 % of the rows at tau_ulps = 8); more -> plain path
        spans = [(lo, min(n, lo + ch)) for lo in range(0, n, ch)]
        # nothing overlaps the host work of the LAST chunk: keep it small (one full-chip round of the kernel)
        tail = 65_536
        if spans[-1][1] - spans[-1][0] >= 2 * tail:
            lo, hi = spans.pop()
            spans += [(lo, hi - tail), (hi - tail, hi)]
        # Every chunk costs the caller's stream a fix-up launch and two dispatch gaps (~30 us): full-size chunks at the
        # head of a long batch are taken two at a time (1M rows: 4 chunks instead of 5, 1.43 against 1.47 ms per step).
        # The last full-size chunk stays single: its ties must be resolved on the host while the two short chunks
        # that end the batch are on the GPU.
        full = [i for i, (lo, hi) in enumerate(spans) if hi - lo == ch]
        if self.pipeline_pair_head and len(spans) >= 5 and len(full) >= 3:
            head = full[:-1]                        # (full-size spans are a prefix of the plan)
            merged = [(spans[head[i]][0], spans[head[i + 1]][1]) for i in range(0, len(head) - 1, 2)]
            if len(head) % 2:
                merged.append(spans[head[-1]])
            spans = merged + spans[len(head):]
            cap = 2 * ch // 32 + 1024
            ch = 2 * ch
        forced = os.environ.get("LSHRS_PLAN")            # experiments: "524288,262144,..." (rows per chunk, must sum to n)
        if forced:
            sizes = [int(v) for v in forced.split(",")]
            if sum(sizes) == n:
                edges = np.cumsum([0] + sizes)
                spans = [(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:])]
                ch = max(sizes)
                cap = ch // 32 + 1024
        return ch, cap, spans

    def _native_pipe(self, lib, dev, cap: int, flag_cap: int):
        """The native pipeline object (lshrs_pipe_*, csrc/pipeline.hip) for this device and these capacities."""
        key = (dev.index, cap, flag_cap)
        pipe = self._pipes.get(key)
        if pipe is None:
            for old in [k for k in self._pipes if k[0] == dev.index and k[1] == cap]:
                lib.lshrs_pipe_destroy(self._pipes.pop(old))      # outgrown stage-1 list
            pipe = lib.lshrs_pipe_create(self.num_bands, self.rows_per_band, self.dim, cap, flag_cap)
            if not pipe:
                raise _native.NativeLibraryError("lshrs_pipe_create failed (out of device or pinned host memory?)")
            self._pipes[key] = pipe
        return pipe

    def _pipelined_native(self, x, out, row_flags, ws, tau, stats, native):
        """The pipelined path driven by the library (csrc/pipeline.hip): same chunks, same kernels and the same host
        engine call per chunk, without the interpreter between the launches, with the tie
        entries and their vectors exported straight into pinned host memory by a kernel on a side stream."""
        t_entry = time.perf_counter()
        torch = _native.require_gpu()
        lib = _native.load()
        dev = x.device
        n = int(x.shape[0])
        eng, planes = native
        timing = self.kernel_events is not None
        # the plan of a batch size is reused: the interpreter's share of a 1.4 ms step is worth trimming
        key = (n, self.pipeline_chunk_rows, int(self._flag_cap_hint), self._projection_version, self.precision,
               self.split_min_rows, self.split_min_elems, self.pipeline_pair_head)
        plan = self._plan_cache.get(key)
        if plan is None:
            ch, cap, spans = self._pipeline_plan(n)
            split = [self._split_applies(hi - lo) for lo, hi in spans]
            flag_cap = max(int(self._flag_cap_hint), ch // 4 + 4096) if any(split) else 0
            nc = len(spans)
            plan = (spans, cap, flag_cap, nc, np.array([0] + [hi for _, hi in spans], dtype=np.int64),
                    np.array(split, dtype=np.uint8), np.zeros(nc, dtype=np.int32), np.zeros(12, dtype=np.int64),
                    np.full(2 * nc, -1.0, dtype=np.float32))
            if len(self._plan_cache) > 64:
                self._plan_cache.clear()
            self._plan_cache[key] = plan
        spans, cap, flag_cap, nc, bounds, chunk_split, status, st, ms = plan    # (status / st / ms: outputs, rewritten per call)
        if torch.cuda.current_device() == dev.index:      # (the device context manager costs ~10 us)
            ctx = contextlib.nullcontext()
        else:
            ctx = torch.cuda.device(dev)
        with ctx:
            pipe = self._native_pipe(lib, dev, cap, flag_cap)
            main = torch.cuda.current_stream(dev)
            rc = lib.lshrs_pipe_hash_f32(
                pipe, x.data_ptr(), x.stride(0), ws.data_ptr(), out.data_ptr(),
                row_flags.data_ptr() if row_flags is not None else None, tau, self._tau1_arg(),
                bounds.ctypes.data, chunk_split.ctypes.data, nc, eng.resolve_fn, eng.handle, planes.ctypes.data,
                status.ctypes.data, ms.ctypes.data if timing else None, st.ctypes.data, main.cuda_stream)
        _native.check(int(rc), "lshrs_pipe_hash_f32")
        sv = st.tolist()
        stats["tie_entries"] += sv[0]
        stats["tie_pairs"] += sv[1]
        for key, i in (("t_head_ms", 3), ("t_enqueue_ms", 4), ("t_wait_ms", 5), ("t_patch_ms", 6), ("t_scatter_ms", 7),
                       ("t_tail_count_ms", 8), ("t_native_ms", 9), ("t_patch_last_ms", 10)):
            stats[key] = sv[i] * 1e-6
        stats["pipeline"] = "native"
        stats["export_topups"] = sv[11]     # chunks whose speculative device->host copy fell short
        if timing:
            for ci, (lo, hi) in enumerate(spans):
                fix = float(ms[2 * ci + 1])
                self.kernel_events.append((float(ms[2 * ci]), None, hi - lo, fix if fix >= 0 else None))
        overflow = []
        if status.any():
            overflow = [spans[ci] for ci in range(nc) if status[ci] != 0]
            if (status == 2).any():
                self._flag_cap_hint = int(sv[2] * 1.25) + 4096
        stats["t_total_ms"] = 1e3 * (time.perf_counter() - t_entry)
        return self._redo_overflowed(x, out, row_flags, overflow, stats)

    def _launch_sig(self, torch, lib, dev, *args, flag_count=None):
        """Enqueue one signature pass (args = the arguments of ``lshrs_sig_hash_batch_f32``).  Returns ``None``, or
        for the split-precision pass ``(flag_count tensor, flag_cap)`` — the caller must compare them once the
        stream has been synchronised and repeat the launch with a larger ``flag_cap`` on overflow."""
        n = int(args[1])
        split = self._split_applies(n)
        flag = None
        if split:
            cap = max(int(self._flag_cap_hint), n // 4 + 4096)
            if self.tau1_ulps > 256.0:
                cap = max(cap, int(n * self.num_bands * self.rows_per_band * min(self.tau1_ulps, 4096.0) * 2.0e-6) + 4096)
            flag_list = torch.empty((cap,), dtype=torch.int64, device=dev)
            if flag_count is None:   # (the pipelined path zeroes one counter per chunk in a single fill)
                flag_count = torch.zeros(1, dtype=torch.int32, device=dev)
            flag = (flag_count, cap, flag_list)
            call = lambda opts: lib.lshrs_sig_hash_batch_split_f32(  # noqa: E731
                *args[:-1], flag_list.data_ptr(), cap, flag_count.data_ptr(), self._tau1_arg(), opts, args[-1])
            name = "lshrs_sig_hash_batch_split_f32"
        else:
            call = lambda opts: lib.lshrs_sig_hash_batch_f32(*args[:-1], opts, args[-1])  # noqa: E731
            name = "lshrs_sig_hash_batch_f32"
        events = self.kernel_events
        if events is None:
            _native.check(call(None), name)
            return flag
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cur = torch.cuda.current_stream(dev)
        mid, opts = None, None
        if split:
            mid = torch.cuda.Event(enable_timing=True)
            mid.record(cur)                      # creates the handle; the library re-arms it on stage 1's dispatch
            opts = _native.SigOpts(events=(None, mid.cuda_event, None, None))
        start.record(cur)
        _native.check(call(ctypes.byref(opts) if opts is not None else None), name)
        end.record(cur)
        events.append((start, end, n, mid))      # split pass: start..mid = stage 1, mid..end = exact fix-up
        return flag

    def _flag_overflow(self, flag) -> bool:
        """After a synchronisation: did the split pass need more room than it had?  (Remembers the need.)"""
        if flag is None:
            return False
        wanted = int(flag[0].item())
        if wanted > flag[1]:
            self._flag_cap_hint = int(wanted * 1.25) + 4096
            return True
        return False

    def _tie_pairs(self, entries: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """Kernel tie entries ``(row*65536 + word, 32-bit column mask)`` -> unique (row, band) pairs,
        sorted by (band, row)."""
        rows = entries[:, 0] >> 16
        words = entries[:, 0] & 0xFFFF
        masks = entries[:, 1].astype(np.uint64)
        band_cols = 8 * self.band_bytes
        codes = []
        if band_cols % 32 == 0:
            # a 32-column word lies inside one band; otherwise (8, 16, 24, 40 ... columns per band)
            # look at the mask bit by bit
            band = (32 * words) // band_cols
            keep = band < self.num_bands
            codes.append(band[keep] * (1 << 48) + rows[keep])
        else:
            for c in range(32):
                hit = ((masks >> np.uint64(c)) & np.uint64(1)).astype(bool)
                if not hit.any():
                    continue
                band = (32 * words[hit] + c) // band_cols
                keep = band < self.num_bands
                codes.append(band[keep] * (1 << 48) + rows[hit][keep])
        code = np.unique(np.concatenate(codes)) if codes else np.empty(0, dtype=np.int64)
        return (code & ((1 << 48) - 1)).astype(np.int64), (code >> 48).astype(np.int32)

    def _tie_engine(self):
        """(engine, planes) when the native host engine is usable for this hasher's shape, else None."""
        eng = None if self.tie_threads == 1 else _hostblas.engine(self.tie_threads)
        if eng is None:
            return None
        cached = self._host_planes_cache
        if cached is None or cached[0] != self._projection_version:
            cached = (self._projection_version,
                      self._stacked().reshape(self.num_bands, self.rows_per_band, self.dim))
            self._host_planes_cache = cached
        return (eng, cached[1]) if eng.shape_trusted(cached[1]) else None

    def _tie_patches(self, xrows: np.ndarray, inverse: np.ndarray, bands: np.ndarray) -> np.ndarray:
        """Band keys of the flagged (row, band) pairs by the reference's own expression
        (``projection @ vector``, ``> 0``, ``np.packbits(..., bitorder='little')``: lsh.py:200-208).

        ``np.matmul(P_band, X[:, :, None])`` runs NumPy's matrix @ vector inner loop once per row, i.e.
        it issues the very ``cblas_sgemv`` call ``P_band @ x`` issues (same operands, same shapes, same
        library) without a Python-level loop; tests/test_tiebreak_host.py checks the two bit for bit.
        One sgemv of this size costs ~0.9 us on one core; with ``tie_threads`` != 1 the pairs go to the
        host engine instead (lshrs_amd/_hostblas.py: the same call from several threads, each through a private
        mapping of NumPy's BLAS, self-checked bit for bit against ``P_band @ x`` per shape).
        ``bands`` arrives sorted, so each band is one contiguous slice.
        """
        m = int(bands.shape[0])
        patch = np.empty((m, self.band_bytes), dtype=np.uint8)
        if m == 0:
            return patch
        native = self._tie_engine()
        if native is not None:
            # the same cblas_sgemv, several cores at once (each worker owns a private mapping of NumPy's BLAS)
            xr = xrows if (xrows.dtype == np.float32 and xrows.ndim == 2 and xrows.strides[1] == 4) \
                else np.ascontiguousarray(xrows, dtype=np.float32)
            return native[0].patch(native[1], xr, inverse, bands)
        planes = self._projections
        starts = np.flatnonzero(np.r_[True, bands[1:] != bands[:-1]])
        stops = np.r_[starts[1:], m]
        for lo, hi in zip(starts, stops):
            plane = np.ascontiguousarray(planes[int(bands[lo])], dtype=np.float32)
            xs = np.ascontiguousarray(xrows[inverse[lo:hi]])
            y = np.matmul(plane, xs[:, :, None])[:, :, 0]
            patch[lo:hi] = np.packbits(y > 0, axis=1, bitorder="little")
        return patch
