"""The device tie replay behind ``LSHHasher`` (mixin): launches of the split pass whose stage 2 replays the host BLAS's order
(``lshrs_sig_hash_batch_split_replay_f32``; stage 2 by key column: ``lshrs_sig_sort``), what comes
back from them (counters, the audit of un-flagged projections, the margin guard), the streaming handle, the live audit against
``P_band @ x`` and the f32 kernel + replay route.  Moved out of ``hasher.py`` in round 5 (no behaviour change): that file keeps the
constructor, the hyperplanes, the windows and the one place that decides which route a batch takes."""

from __future__ import annotations

import contextlib
import ctypes
import time

import numpy as np

from . import _native
from .windows import bound_tau1_ulps, escalated_window


class _RawStreamWait:
    """What a synchronous launch waits on: the stream it was enqueued on, by its raw handle (`lshrs_stream_synchronize`)."""

    __slots__ = ("_lib", "_raw")

    def __init__(self, lib, raw) -> None:
        self._lib, self._raw = lib, raw

    def synchronize(self) -> None:
        _native.check(self._lib.lshrs_stream_synchronize(self._raw), "lshrs_stream_synchronize")

    def query(self) -> bool:
        return False


class _EpochWait:
    """What a synchronous launch of the split pass waits on: the `done` word of its counter block (`lshrs_wait_done`: a bounded
    poll of pinned memory - the runtime's stream wait wakes 10 - 20 us late - then the stream)."""

    __slots__ = ("_lib", "_word", "_epoch", "_spin_ns", "_raw")

    def __init__(self, lib, word, epoch, spin_ns, raw) -> None:
        self._lib, self._word, self._epoch, self._spin_ns, self._raw = lib, word, epoch, spin_ns, raw

    def synchronize(self) -> None:
        rc = self._lib.lshrs_wait_done(self._word, self._epoch, self._spin_ns, self._raw)
        if rc != 0:
            _native.check(rc, "lshrs_wait_done")

    def query(self) -> bool:
        return False


class _PendingKeys:
    """Handle of :meth:`LSHHasher.hash_device_async`."""

    def __init__(self, hasher: "LSHHasher", x, out, row_flags, state) -> None:
        self._hasher, self._x, self._out, self._row_flags, self._state = hasher, x, out, row_flags, state
        self._stats = dict(hasher.last_stats) if state is None else None     # (a handle that was complete on creation)
        # the stream the launch was enqueued on: whoever verifies this handle - it may be another thread that needs a counter
        # block - repeats it THERE if it must be repeated (the buffers are ordered against that stream, not the verifier's)
        self._stream = _native.require_gpu().cuda.current_stream(x.device) if state is not None else None

    def done(self) -> bool:
        return self._state is None or bool(self._state[0].query())

    def _finish_locked(self):
        """Verify the launch (repeat it where it must be repeated); returns THIS batch's statistics - `last_stats` of the
        hasher may already be another batch's by the time the caller looks (a repeat finishes what is pending)."""
        h = self._hasher
        if self in h._async_pending:
            h._async_pending.remove(self)
        state, self._state = self._state, None
        if state is None:
            return self._stats
        stats = {"n": int(self._x.shape[0]), "tie_entries": 0, "tie_pairs": 0, "relaunches": 0}
        if not h._replay_finish(state, stats):        # the stage-1 list was too small: once more, synchronously, with room
            with _native.require_gpu().cuda.stream(self._stream):
                h._hash_device_locked(self._x, self._out, self._row_flags, "host", host_rows=None)
            stats["relaunches"] += h.last_stats.get("relaunches", 0)
            for k in ("tie_entries", "tie_pairs", "tie_break_engine"):
                if k in h.last_stats:
                    stats[k] = h.last_stats[k]
        h.last_stats = stats
        self._stats = stats
        self._x = None
        return stats

    def result(self):
        """The ``(n, num_bands, band_bytes)`` uint8 keys tensor, final and verified."""
        if self._state is not None:
            with self._hasher._lock:
                self._finish_locked()
        return self._out

    @property
    def stats(self) -> dict:
        """THIS batch's statistics (route, flagged projections, ties, relaunches, the audit ...), once it is verified - the
        per-call form of ``LSHHasher.last_stats``, which is the hasher's LAST batch and may be another thread's."""
        self.result()
        return dict(self._stats or {})



class _ReplayPaths:
    def _replay_launch(self, x, out, row_flags, ws, tau, model, want_event: bool = False):
        """Enqueue one split pass with the tie replay on the current stream; returns what `_replay_finish` needs.
        The counters of the launch come back through one of four pinned blocks, taken from a free list and returned by
        `_replay_finish` (streamed launches keep at most three; a synchronous caller that takes the last one keeps the lock
        until it is back).  Scratch is per (device, stream): launches enqueued on different streams never share a list or a
        counter block; launches on one stream are ordered by the stream.
        Round 6 (the lean step): everything a launch passes that does not change from call to call - the option structs of
        the four counter blocks with their `done` words, the audit struct, their ctypes references - is made once per
        scratch; a synchronous caller waits on the `done` word of its block (`lshrs_wait_done`: a bounded poll, then the
        stream) instead of sleeping on the stream."""
        torch = _native.require_gpu()
        lib = _native.load()
        dev = x.device
        n = int(x.shape[0])
        timing = self.kernel_events is not None
        switch = torch._C._cuda_getDevice() != dev.index
        if switch:
            guard = torch.cuda.device(dev)
            guard.__enter__()
        try:
            # (the raw handle of the device's current stream: building a torch Stream object per launch costs more than the
            #  arithmetic around it; the object is made where something needs it - events, the error path)
            raw = torch._C._cuda_getCurrentRawStream(dev.index)
            cap = max(int(self._flag_cap_hint), n // 4 + 4096)
            if self.tau1_ulps > 256.0:      # a wide (e.g. "bound") window flags ~1.3e-6 of the projections per unit
                cap = max(cap, int(n * self.num_bands * self.rows_per_band * min(self.tau1_ulps, 4096.0) * 2.0e-6) + 4096)
            skey = (dev.index, raw)
            scratch = self._replay_scratch.get(skey)
            if scratch is None or scratch[9][3] < cap:
                nc = _native.SIG_COUNTERS
                pinned = torch.zeros((4, nc + 8), dtype=torch.int32).pin_memory()       # per block: the counters, then its `done` word
                audit = _native.SigAudit(0, 0, 0, 0, 0)
                scratch = (torch.empty((cap,), dtype=torch.int64, device=dev),
                           torch.zeros(_native.SIG_DEVICE_COUNTERS, dtype=torch.int32, device=dev),   # counters + stage-2 slots
                           pinned, pinned.numpy(), [0, [3, 2, 1, 0]],             # launches so far, free pinned blocks
                           torch.empty((cap,), dtype=torch.float32, device=dev),   # stage-1 value of every list entry
                           # the audit sample of a launch: entry, (stage-1 value, window) per slot
                           torch.empty((2 * max(1, self.audit_unflagged),), dtype=torch.int64, device=dev),
                           torch.empty((4 * max(1, self.audit_unflagged),), dtype=torch.float32, device=dev), [audit])
                audit.list, audit.vals, audit.slots = scratch[6].data_ptr(), scratch[7].data_ptr(), int(scratch[6].shape[0])
                opts4 = [_native.SigOpts() for _ in range(4)]
                for i, o in enumerate(opts4):
                    o.done_host = pinned.data_ptr() + 4 * ((nc + 8) * i + nc)
                # ... and what every launch passes of it: list / counter / value pointers, the list's capacity, the four pinned
                # blocks, the option struct of each block with its reference, the audit struct's reference, the epoch counter
                scratch += ((scratch[0].data_ptr(), scratch[1].data_ptr(), scratch[5].data_ptr(), int(scratch[0].shape[0]),
                             tuple(pinned.data_ptr() + 4 * (nc + 8) * i for i in range(4)), opts4,
                             tuple(ctypes.byref(o) for o in opts4), ctypes.byref(audit), [0],
                             tuple(o.done_host for o in opts4)),)
                if len(self._replay_scratch) >= 16 and not self._async_pending:
                    # a caller that keeps making new streams must not pile up lists: nothing is in flight, start over
                    self._replay_scratch.clear()
                    self._replay_events.clear()
                    self._sort_res.clear()
                self._replay_scratch[skey] = scratch
            # (the device counters are zero: at creation, and the launch that exports them leaves them so)
            counts, host_counts, turn, audit_box, ptrs = scratch[1], scratch[3], scratch[4], scratch[8], scratch[9]
            while not turn[1]:        # every pinned block belongs to an unverified launch: verify the oldest streamed one
                if not self._async_pending:      # (cannot happen: a synchronous caller that takes the last block keeps the lock)
                    raise _native.NativeLibraryError("no free counter block for a replay launch")
                self._async_pending[0]._finish_locked()
            slot = turn[1].pop()
            turn[0] += 1
            opts = ptrs[5][slot]
            epoch = ptrs[8][0] = (ptrs[8][0] % 0x7FFFFFF0) + 1
            opts.done_epoch = epoch
            ev = None
            if timing:
                # four timing events per launch, from a ring as deep as the pinned blocks (creating and recording them
                # afresh costs ~20 us of host time per launch - on a 1.2 ms step that is the measurement disturbing
                # the measured); they travel in the call's own lshrs_sig_opts
                ring = self._replay_events.get(skey)
                if ring is None:
                    cur = torch.cuda.current_stream(dev)
                    ring = []
                    for _ in range(4):
                        quad = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                        for e in quad:
                            e.record(cur)            # creates the handles; the library re-arms them on its dispatches
                        ring.append(quad)
                    self._replay_events[skey] = ring
                ev = ring[slot]
                (opts.ev_stage1_start, opts.ev_stage1_stop, opts.ev_stage2_start, opts.ev_stage2_stop) = [e.cuda_event for e in ev]
            elif opts.ev_stage1_start:
                opts.ev_stage1_start = opts.ev_stage1_stop = opts.ev_stage2_start = opts.ev_stage2_stop = None
            sort = None
            if self._stage2_mode() is not None:
                sort = self._sort_scratch(torch, dev, skey, cap)
                sort[0].parity = sort[4][0] & 1              # the set of column counters this launch counts in (the other: cleared by it)
                sort[4][0] += 1
                if opts._sort_ref is not sort[0]:
                    opts.set_sort(sort[0])
            elif opts.sort:
                opts.set_sort(None)
            audit_ref = None
            if self.audit_unflagged > 0:
                self._audit_seed = (self._audit_seed + 1) & 0x7FFFFFFF
                audit = audit_box[0]                 # (the struct of this scratch: the library reads it during the call only)
                audit.target = self.audit_unflagged
                audit.seed = (self._audit_seed * 2654435761) & 0xFFFFFFFF
                audit_ref = ptrs[7]
            rc = lib.lshrs_sig_hash_batch_split_replay_f32(
                x.data_ptr(), n, x.stride(0), ws.data_ptr(), self.num_bands, self.rows_per_band, self.dim,
                out.data_ptr(), ptrs[1], tau, row_flags.data_ptr() if row_flags is not None else None,
                ptrs[0], ptrs[2], ptrs[3], self._tau1_arg(), model, ptrs[4][slot], audit_ref, ptrs[6][slot], raw)
            if rc != 0:
                torch.cuda.current_stream(dev).synchronize()
                counts.zero_()              # a failed launch may have left counts behind: the next call starts from zero
                if sort is not None:
                    sort[3][2].zero_()
                turn[1].append(slot)
                _native.check(rc, "lshrs_sig_hash_batch_split_replay_f32")
            if want_event:          # (the streaming handle asks the event whether the launch is done)
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(dev))
            else:
                done = _EpochWait(lib, ptrs[9][slot], epoch, int(self.spin_wait_us) * 1000, raw)
        finally:
            if switch:
                guard.__exit__(None, None, None)
        # (the ceiling the live check holds this launch to is the one of the coefficients it was launched with: `window_info`
        #  follows whichever BLAS-order model `_ensure_window` set last, and an async handle may be finished after a switch)
        return (done, host_counts[slot:slot + 1, :_native.SIG_COUNTERS], slot, (ptrs[3],), n, ev,
                float("inf") if self.window_mode["tau1"] == "bound" else float(self.tau1_ulps),
                float(self.window_info.get("window_units_worst_case_row", float("inf"))), turn[0], turn)

    def _stage2_mode(self):
        """Which stage 2 a launch of the split pass asks for: 1 = by key column, through the buckets stage 1 fills; None = the
        plain stage 2 (what the resident-image kernel's short rows always take)."""
        want = self.stage2_sorted
        if want not in ("auto", "buckets", False):
            raise ValueError("stage2_sorted must be 'auto', 'buckets' or False")
        if want is False or self.dim <= 128 or self.rows_per_band == 1:      # (one-row bands: stage 2 is the sdot replay, one list)
            return None
        padcols = self.num_bands * self.band_bytes * 8
        if padcols > _native.SORT_MAX_COLS:
            return None
        if self._resident_shape():    # (the resident-image kernel keeps one list: its stage 2 is the plain one)
            return None
        return 1                      # "auto", "buckets"

    def _sort_scratch(self, torch, dev, skey, cap: int):
        """(SigSort, a SigOpts that carries it, capacity, tensors, launch counter) for launches on this (device, stream): a
        segment per padded key column - 1.5 x its share of the list + 64 entries -, the entries' stage-1 values and the audit
        entries' windows beside them, two sets of 1024 column counters."""
        padcols = self.num_bands * self.band_bytes * 8
        per = max(int(1.5 * cap / padcols) + 64, int(self._bucket_cap_hint))
        need = per * padcols
        got = self._sort_res.get(skey)
        if got is None or got[2] < need:
            lst = torch.empty(need, dtype=torch.int64, device=dev)
            y = torch.empty(need, dtype=torch.float32, device=dev)
            hist = torch.zeros(2 * _native.SORT_MAX_COLS, dtype=torch.int32, device=dev)
            thr = torch.empty(need, dtype=torch.float32, device=dev)
            sort = _native.SigSort(lst.data_ptr(), y.data_ptr(), hist.data_ptr(), need, thr.data_ptr(), 1)
            got = (sort, _native.SigOpts(sort=sort), need, (lst, y, hist, thr), [0])
            self._sort_res[skey] = got
        return got

    def _replay_finish(self, state, stats) -> bool:
        """Wait for a launch of `_replay_launch`; False when it must be repeated: its stage-1 list was too small (more
        room next time), or the stage-1 deviation measured on its flagged projections came within `margin_guard` of
        the window (the hasher switches to the deterministic bound and stays there)."""
        done, host_counts, slot, caps, n, ev, window, worst = state[:8]
        try:
            done.synchronize()      # (the launch behind stage 2 has written the counters into the pinned block)
        finally:
            state[9][1].append(slot)        # (the pinned block is free for the next launch - also when the wait raised)
        ties, flagged, _, flips, audited, audit_bad, _, col_over = host_counts[0].tolist()
        as_float = host_counts[0].view(np.float32)
        max_dev, audit_ratio = float(as_float[2]), float(as_float[6])
        over = flagged > caps[0]
        if col_over:        # buckets: one key column wanted more than its segment holds (rows aligned with a hyperplane): double them
            over = True
            per = max(64, int(self._bucket_cap_hint), int(1.5 * caps[0] / max(1, self.num_bands * self.band_bytes * 8)) + 64)
            self._bucket_cap_hint = 2 * per
            flagged = max(flagged, caps[0])
        if over:
            self._flag_cap_hint = int(flagged * 1.25) + 4096      # (rows flagged wholesale: NaN / Inf / extreme scales)
            stats["relaunches"] += 1
            return False
        stats["max_dev_units"] = max(max_dev, stats.get("max_dev_units", 0.0))
        # the audit of what stage 1 did NOT flag: whatever the window mode, a sign that is not the host's is a wrong key bit
        stats["audited_unflagged"] = stats.get("audited_unflagged", 0) + audited
        stats["audit_sign_disagreements"] = stats.get("audit_sign_disagreements", 0) + audit_bad
        stats["audit_max_window_ratio"] = max(audit_ratio, stats.get("audit_max_window_ratio", 0.0))
        tot = self.audit_totals
        tot["audited"] += audited
        tot["sign_disagreements"] += audit_bad
        tot["max_window_ratio"] = max(tot["max_window_ratio"], audit_ratio)
        if (audit_bad or audit_ratio > 1.0) and window != float("inf"):
            # a MEASURED window refuted on this batch's own data (rows like tests/_adversary.py get past the margin guard,
            # which only sees flagged projections): the proven window from here on, and this batch once more
            if self.window_mode["tau1"] != "bound":
                self.tau1_ulps, self.window_mode["tau1"] = bound_tau1_ulps(self.dim), "bound"
                self._window_set.clear()
            self.margin_escalations += 1
            self.audit_escalations = getattr(self, "audit_escalations", 0) + 1
            stats["relaunches"] += 1
            stats["audit_escalations"] = self.audit_escalations
            stats["margin_escalations"] = self.margin_escalations
            return False
        if audit_bad or audit_ratio > 1.0:
            raise _native.NativeLibraryError(
                f"audit of the projections stage 1 decided on its own: {audit_bad} of {audited} sampled key bits are not the "
                f"sign of the host's value, largest |y_stage1 - y_host| / window = {audit_ratio:.3f} (> 1 means outside the "
                f"window): the stage-1 window ({'proven' if window == float('inf') else f'{window:.0f} units, measured'}) "
                "does not cover this data on this device - keys of this batch are not the reference's")
        if window == float("inf"):
            # proven window: what stage 2 measured on every flagged projection can only be INSIDE it - anything else is a
            # bug in the bound or in the arithmetic model it rests on, and must not pass silently
            if max_dev > worst:
                raise _native.NativeLibraryError(
                    f"stage 1 strayed {max_dev:.1f} units from the host's value, outside the proven window "
                    f"({worst:.1f} units at most): the window's premises do not hold on this device")
        elif (self.margin_guard > 0.0 and max_dev > self.margin_guard * window
                and window < bound_tau1_ulps(self.dim)):
            # the measured margin of this batch is not what the window IT WAS LAUNCHED WITH assumes (another batch in
            # flight may have widened the hasher's window since): hash it again, and only ever widen the hasher's window
            wider, mode = escalated_window(window, max_dev, self.dim)
            if self.window_mode["tau1"] != "bound" and wider > self.tau1_ulps:
                self.tau1_ulps, self.window_mode["tau1"] = wider, mode      # ("bound": the proven window from here on)
                if mode == "bound":
                    self._window_set.clear()
            self.margin_escalations += 1
            stats["relaunches"] += 1
            stats["margin_escalations"] = self.margin_escalations
            return False
        if ev is not None and self.kernel_events is not None:
            self.kernel_events.append((ev[0].elapsed_time(ev[1]), None, n, ev[2].elapsed_time(ev[3])))
        stats["tie_entries"] = ties
        stats["tie_pairs"] = ties          # (tied PROJECTIONS here: each decided by the replayed host order)
        stats["flagged"] = flagged         # projections inside the stage-1 window: every one decided by stage 2
        stats["sign_flips"] = flips        # ... of which stage 1 had the sign wrong
        stats["tau1_ulps"] = self.tau1_ulps if window == float("inf") else window
        stats["window"] = "proven" if window == float("inf") else "measured"
        stats["margin_escalations"] = self.margin_escalations
        stats["tie_break_engine"] = "device-replay"
        return True

    def _hash_device_replay(self, x, out, row_flags, ws, tau, stats, model, yield_lock: bool = False):
        """One launch of the split pass whose stage 2 also breaks the ties (``lshrs_sig_hash_batch_split_replay_f32``):
        the keys are the reference's when the stream has run.  The only host step is reading two counters back
        (stage-1 list overflow -> repeat with room; tied projections -> stats)."""
        while self._async_pending:      # (their pinned pairs are handed out in turn: verify them before taking more)
            self._async_pending[0]._finish_locked()
        while True:
            state = self._replay_launch(x, out, row_flags, ws, tau, model)
            if yield_lock and state[9][1]:      # (whoever takes the LAST pinned block keeps the lock until it is free again)
                # (only for `hash_device` itself - the callers that drive shared staging buffers keep the lock.)  The
                # hasher's lock covers what is SHARED - the hand-out of scratch, pinned counter blocks and turns, the window
                # switch - not the wait for the device: another thread (another stream) may enqueue its batch meanwhile.
                # Launches on one stream are ordered by the stream; every launch has a pinned counter block of its own.
                self._lock.release()
                try:
                    state[0].synchronize()
                finally:
                    self._lock.acquire()
            if self._replay_finish(state, stats):
                break
            self._ensure_window(x.device, ws, model)      # (a guard that has just moved the hasher to the proven window)
        undisturbed = state[9][0] == state[8]         # nobody has launched over this launch's list since
        if self.reference_blas != "host":
            stats["reference_blas"] = self.reference_blas
        if self.audit_every > 0 and stats.get("flagged", 0) > 0 and undisturbed and self._host_blas_agrees():
            self._audit_countdown -= 1
            if self._audit_countdown <= 0 and (time.monotonic() - self._audit_last >= self.audit_min_interval_s
                                               or self._audit_last == 0.0):
                self._audit_countdown = self.audit_every
                self._audit_last = time.monotonic()
                if not self._audit_replay(x, out, stats):
                    # what the device decided is not what this process's NumPy computes: the replay's licence is void
                    # for this hasher - the host engine (the library's own call) takes over, starting with this batch
                    self.tie_replay = "off"
                    self.audit_failures += 1
                    out = self._hash_device_locked(x, out, row_flags, "host", host_rows=None)
                    self.last_stats["audit_failures"] = self.audit_failures
                    return out
        self.last_stats = stats         # (the lock was yielded during the wait: another thread's batch may have put its own here)
        return out

    def _audit_replay(self, x, out, stats, sample: int = 16) -> bool:
        """Spot check of the device's decisions against the reference's own expression on live data: a handful of the
        projections stage 2 has just decided are re-evaluated with ``P_band @ x`` (lshrs/hash/lsh.py:200) on the host
        and compared with the key bits.  The BLAS-order model is licensed on synthetic vectors at first use and when the
        BLAS's configuration changes; this closes the loop on real inputs, every ``audit_every`` batches (a few small
        device ops and one copy)."""
        torch = _native.require_gpu()
        dev = x.device
        scratch = self._replay_scratch.get((dev.index, torch.cuda.current_stream(dev).cuda_stream))
        if scratch is None:
            return True
        k = min(sample, int(stats["flagged"]), int(scratch[0].shape[0]))
        if k <= 0:
            return True
        # list entries, their rows of x and of the keys: gathered on the device, ONE copy to the host
        items_d = scratch[0][:k]
        sort = self._sort_res.get((dev.index, torch.cuda.current_stream(dev).cuda_stream))
        if sort is not None and sort[0].mode == 1 and self._stage2_mode() == 1:
            # buckets: the flagged projections sit in their key columns' segments - the first entry of up to `sample` columns
            # that have one (the counters of the pass that has just run are the set it was not told to clear)
            lst, _, hist, _ = sort[3]
            padcols = self.num_bands * self.band_bytes * 8
            per = int(sort[0].cap) // padcols
            used = ((sort[4][0] - 1) & 1) * _native.SORT_MAX_COLS
            cols = torch.nonzero(hist[used:used + padcols] > 0)[:sample, 0]
            if cols.numel() == 0:
                return True
            items_d = lst[cols * per] & ~(1 << 62)            # (an audit entry is a projection like any other here)
            k = int(items_d.shape[0])
        rows_d = (items_d >> 21).clamp_(0, int(x.shape[0]) - 1)
        nx, nk = k * self.dim * 4, k * self.num_bands * self.band_bytes
        packed = torch.cat([items_d.view(torch.uint8), x.index_select(0, rows_d).reshape(-1).view(torch.uint8),
                            out.index_select(0, rows_d).reshape(-1)]).cpu().numpy()
        items = packed[:8 * k].view(np.int64)
        xr = packed[8 * k:8 * k + nx].view(np.float32).reshape(k, self.dim)
        kb = packed[8 * k + nx:8 * k + nx + nk].reshape(k, self.num_bands, self.band_bytes)
        rows, cols = items >> 21, (items & ((1 << 21) - 1)).astype(np.int64)
        band_cols = 8 * self.band_bytes
        keep = (cols // band_cols < self.num_bands) & (cols % band_cols < self.rows_per_band) & (rows < x.shape[0])
        rows, cols, xr, kb = rows[keep], cols[keep], xr[keep], kb[keep]
        if rows.size == 0:
            return True
        stats["audited"] = stats.get("audited", 0) + int(rows.size)
        for i in range(rows.size):
            band, bit = int(cols[i] // band_cols), int(cols[i] % band_cols)
            y = np.ascontiguousarray(self._projections[band], dtype=np.float32) @ np.ascontiguousarray(xr[i])
            want = bool(y[bit] > 0)
            have = bool((kb[i, band, bit >> 3] >> (bit & 7)) & 1)
            if want != have:
                return False
        return True

    def hash_device_async(self, x, *, out=None, row_flags=None):
        """:meth:`hash_device` for streaming ingest: enqueue the batch and return a handle at once; ``handle.result()``
        returns the keys once the launch has been VERIFIED (the one thing the host must look at - whether the stage-1
        list held - and the repeat with room if it did not), so the host's wake-up and the interpreter overlap the next
        batch's kernels instead of idling the GPU between batches (~55 us per 1M x 768 batch).  At most three batches
        stay unverified: a fourth call verifies the oldest first.  Where the device tie replay does not apply the
        batch is hashed synchronously and the handle is complete on return."""
        torch = _native.require_gpu()
        if x.dim() != 2 or x.shape[1] != self.dim:
            raise ValueError(f"Expected vectors of dimension {self.dim}, received {tuple(x.shape)}")
        if x.dtype != torch.float32 or not x.is_cuda:
            raise TypeError("hash_device expects a float32 CUDA/ROCm tensor")
        if x.stride(1) != 1:
            x = x.contiguous()
        with self._lock:
            return self._hash_device_async_locked(x, out, row_flags)

    def _hash_device_async_locked(self, x, out, row_flags):
        torch = _native.require_gpu()
        n = int(x.shape[0])
        while len(self._async_pending) >= 3:
            self._async_pending[0]._finish_locked()
        model = 0
        if (n > 0 and self.tie_break == "host" and self.tie_replay == "auto" and self._split_applies(n, replay=True)
                and x.stride(0) < (1 << 20)):          # (round 5: rows at any 4-byte address)
            model = self._replay_model()
        if not model or model == 3:          # (model 3: at most eight elements on the SkylakeX build - the plain-load replay, synchronous)
            return _PendingKeys(self, x, self._hash_device_locked(x, out, row_flags, self.tie_break, host_rows=None),
                                row_flags, None)
        bb = self.band_bytes
        if out is None:
            out = torch.empty((n, self.num_bands, bb), dtype=torch.uint8, device=x.device)
        elif out.shape != (n, self.num_bands, bb) or out.dtype != torch.uint8 or not out.is_contiguous():
            raise ValueError("out must be a contiguous uint8 tensor of shape (n, num_bands, band_bytes)")
        ws = self._workspace(x.device)
        self._ensure_window(x.device, ws, model)
        state = self._replay_launch(x, out, row_flags, ws, self._tau_arg(), model, want_event=True)
        handle = _PendingKeys(self, x, out, row_flags, state)
        self._async_pending.append(handle)
        return handle

    def _hash_device_f32_replay(self, x, out, row_flags, ws, tau, stats, model):
        """The exact-f32 kernel followed by the device's tie replay (``lshrs_sig_resolve_ties_replay_f32``): for batches
        and shapes the split pass does not take (fewer than 256 key columns, fewer than 256 rows).  Same bytes, no
        host arithmetic; the host reads two counters back (lists too small -> repeat with room)."""
        torch = _native.require_gpu()
        lib = _native.load()
        dev = x.device
        n = int(x.shape[0])
        ctx = contextlib.nullcontext() if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)
        with ctx:
            cur = torch.cuda.current_stream(dev)
            flags_ptr = row_flags.data_ptr() if row_flags is not None else None
            cap = min(max(4096, n // 16 + 4096), 2 ** 30)
            fcap = 2 * cap
            while True:
                key = ("f32", dev.index, cur.cuda_stream)
                scratch = self._replay_scratch.get(key)
                if scratch is None or scratch[0].shape[0] < cap or scratch[1].shape[0] < fcap:
                    pinned = torch.zeros(_native.SIG_COUNTERS, dtype=torch.int32).pin_memory()
                    scratch = (torch.empty((cap, 2), dtype=torch.int64, device=dev),
                               torch.empty((fcap,), dtype=torch.int64, device=dev),
                               torch.zeros(_native.SIG_DEVICE_COUNTERS, dtype=torch.int32, device=dev), pinned, pinned.numpy())
                    self._replay_scratch[key] = scratch
                tie_list, flag_list, counts, pinned, host_counts = scratch
                tcap, lcap = int(tie_list.shape[0]), int(flag_list.shape[0])
                cptr = counts.data_ptr()
                _native.check(
                    lib.lshrs_sig_hash_batch_f32(x.data_ptr(), n, x.stride(0), ws.data_ptr(), self.num_bands,
                                                 self.rows_per_band, self.dim, out.data_ptr(), tie_list.data_ptr(), tcap,
                                                 cptr, tau, flags_ptr, None, cur.cuda_stream),
                    "lshrs_sig_hash_batch_f32")
                if n < 256:
                    # a query vector or a handful: almost never a tie (2.6 per 1000 vectors) - look at the count before
                    # spending three more launches on an empty list
                    wanted = int(counts[0:1].item())
                    if wanted == 0:
                        items = 0
                        break
                    if wanted > tcap:       # (a tiny batch of pathological rows: start over with room, counters zeroed)
                        counts.zero_()
                        cap = max(cap, wanted)
                        stats["relaunches"] += 1
                        continue
                _native.check(
                    lib.lshrs_sig_resolve_ties_replay_f32(x.data_ptr(), n, x.stride(0), ws.data_ptr(), self.num_bands,
                                                          self.rows_per_band, self.dim, out.data_ptr(),
                                                          tie_list.data_ptr(), tcap, cptr, tau, flag_list.data_ptr(),
                                                          lcap, model, pinned.data_ptr(), cur.cuda_stream),
                    "lshrs_sig_resolve_ties_replay_f32")
                cur.synchronize()
                wanted, items = int(host_counts[0]), int(host_counts[1])
                if wanted <= tcap and items <= lcap:
                    break
                cap, fcap = max(cap, wanted), max(fcap, 2 * wanted, items)      # the kernels counted what they wanted to write
                stats["relaunches"] += 1
        stats["tie_entries"] = wanted
        stats["tie_pairs"] = items          # (tied projections)
        stats["tie_break_engine"] = "device-replay"
        return out
