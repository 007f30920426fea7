"""``LSHHasher`` — the signature pass of lshrs on MI355X.

Same constructor, attributes, methods and error messages as the reference class
(lshrs/hash/lsh.py:51-247); the arithmetic runs in ``csrc/sig16.hip`` / ``sig16r.hip`` / ``sig_replay.hip`` / ``sig_f32.hip`` / ``sig_small.hip`` through
the C ABI of ``include/lshrs_hip.h``.  There is no CPU hashing path in here.

Bit-exactness.  The reference's bits are ``sign(sgemv_f32(P_band, x))`` as rounded by the *host's* BLAS.
Every route here is a fast first evaluation, a WINDOW that contains every projection whose sign could differ from the
host's, and an exact decision for what is inside it - the host BLAS's own value, replayed on the device in the library's
summation order (licensed by a bit-for-bit check against this process's NumPy) or, where that order is not recognised,
computed by the library itself on the host.  The windows are PROVEN by default (``lshrs_amd/windows.py``): nothing is left
to the first evaluation that could come out differently on the host.  ``tie_break="none"`` returns the raw kernel bits;
``last_stats`` says which route a batch took and how many projections were decided exactly.
"""

from __future__ import annotations

import contextlib
import ctypes
import threading
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _hostblas, _native
from ._config import HashSignatures  # noqa: F401  (re-exported)

__all__ = ["LSHHasher", "HostBlasNotRecognised"]


class HostBlasNotRecognised(UserWarning):
    """This process's BLAS sums `P_band @ x` in an order the device replay does not know (an OpenBLAS / NumPy this library
    was not verified against): keys stay the host's own - every near-tie is decided by the host engine calling that BLAS -
    at the host engine's rate (~20 M vectors/s instead of ~800 M).  Naming the build (`reference_blas="openblas-skylakex"` /
    `"openblas-haswell"`) keeps the device route and pins the keys to that build instead."""

from ._host_engine_route import _HostEngineRoute
from ._host_paths import _HostPaths, _OneRequest  # noqa: F401
from ._replay_paths import _PendingKeys, _RawStreamWait, _ReplayPaths  # noqa: F401
from .windows import (MFMA_BF16_ERR_UNITS, _U, _bf16_rne, bound_tau1_ulps, bound_tau_ulps, default_tau1_ulps,  # noqa: F401
                      escalated_window, window_coefficients)


class _ProjectionList(list):
    """``list`` of per-band hyperplane matrices that notices item re-assignment."""

    def __init__(self, items, owner: "LSHHasher") -> None:
        super().__init__(items)
        self._owner = owner

    def _touch(self) -> None:
        self._owner._projection_version += 1

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        self._touch()

    def __delitem__(self, key):
        super().__delitem__(key)
        self._touch()

    def append(self, value):
        super().append(value)
        self._touch()

    def extend(self, values):
        super().extend(values)
        self._touch()

    def insert(self, index, value):
        super().insert(index, value)
        self._touch()

    def pop(self, *a):
        out = super().pop(*a)
        self._touch()
        return out

    def clear(self):
        super().clear()
        self._touch()

    def __reduce__(self):  # pickle as a plain list (LSHRS.__getstate__ reads this attribute)
        return (list, (list(self),))



class LSHHasher(_HostPaths, _HostEngineRoute, _ReplayPaths):
    """Sign-random-projection hasher; drop-in for ``lshrs.hash.lsh.LSHHasher``.

    Extra keyword arguments (not in the reference):
      device      torch device index / ``torch.device`` (default: current device at call time)
      tie_break   "host" (default: bytes equal the reference on this host), or "none" (raw kernel bits)
      tie_threads host tie-break workers: None = auto (this process's share of the cores, at most 8), 1 = NumPy only
      precision   "bf16x3" (default): batches whose shape allows it (dim % 4 == 0, >= 256 key columns - or 128 .. 224
                  with dim >= 384 -, hyperplane norms in [2^-40, 2^40]) take the split-precision first pass - bf16 matrix cores,
                  then the exact decision for every projection inside the stage-1 window - everything else the f32
                  kernel; the keys are the same either way.  "f32": always the f32 kernel.
      tau1_ulps   stage-1 window of the split pass.  Default (None, or "bound"): the PROVEN window - per-hyperplane
                  coefficients (``windows.window_coefficients``) times the norms of the two bf16 pieces of the row, which
                  stage 1 measures: 339 units of 2^-24 ||x|| ||p|| on Gaussian rows at 768-d, more for a row whose
                  residual is larger (``window_info``, ``last_stats["tau1_ulps"]``).  Stage 2 still measures
                  |y_stage1 - y_hostBLAS| on every flagged projection (``last_stats["max_dev_units"]``): a value outside the
                  window raises.  A number: a window of that many units - round 2's default was 64 x sqrt(768 / dim), 3-4x
                  the largest deviation seen on any input family, watched by ``margin_guard``; it is ~9 % faster and a
                  statistical statement: rows built for the purpose (tests/_adversary.py) get past it AND past the guard.
      tau_ulps    tie window of the f32 kernel's fmaf chain against the host, same convention: proven by default (281 units
                  at 768-d), a number for a measured one (round 2: 8).
      margin_guard  (numeric windows only) fraction of the window the measured deviation may reach before the batch is hashed
                  again with a wider one - up to the proven window (default 0.5; 0 disables the guard)
      audit_unflagged  projections per launch of the split pass that stage 1 decided ON ITS OWN (did not flag) and that stage
                  2 verifies anyway - a pseudo-random sample, different in every launch: the replayed host-BLAS sign against the
                  key bit stage 1 stored, and |y_stage1 - y_hostBLAS| against the window that projection was compared with
                  (default 4096: < 0.5 % of a 1M x 768 step; 0 = off).  ``last_stats["audited_unflagged"]``, ``["audit_sign_disagreements"]``,
                  ``["audit_max_window_ratio"]``; a disagreement or a ratio above 1 means a key bit the reference would not have
                  produced: with the proven window it raises (the window's premises - the instruction model - do not hold on
                  this device), a measured window (``tau1_ulps=<number>``) is replaced by the proven one and the batch repeated
      audit_every every n-th synchronous ``hash_device`` batch (default 64, and the first) a handful of the projections
                  stage 2 decided are re-evaluated with ``P_band @ x`` on the host and compared with the key bits; a
                  disagreement revokes the device replay for this hasher (``audit_failures``) and the
                  batch is hashed again with the host engine.  0 = never
      tie_replay  "auto" (default): batches that take the split pass break their ties on the device (stage 2 replays
                  the host BLAS's summation order, recognised and verified at first use); "off": host engine only
      reference_blas  which BLAS the keys are the reference's keys ON.  The reference's bits are whatever `projection @ vector`
                  (lsh.py:200) returns on the machine that runs it - its BLAS's summation order decides the projections that
                  are ties.  "host" (default): this process's NumPy, recognised and verified at first use (a host whose
                  BLAS is not recognised hashes through the host engine: same keys, 20 M instead of 800 M vectors/s).
                  "openblas-skylakex" / "openblas-haswell": the order of that build of OpenBLAS 0.3.2x (what NumPy's
                  wheels ship) replayed on the device WHATEVER BLAS this host has - an index and its queries hash alike on
                  every machine that names the same build, and no host loses the device path.  SkylakeX = every CPU with
                  AVX-512 (Intel servers since Skylake-X, AMD Zen 4 / Zen 5 - EPYC 9004 / 9005); Haswell = AVX2 only (Intel
                  Haswell .., AMD Zen 1 - 3); `LSHHasher.host_blas_name()` says which one THIS process's NumPy runs.
                  The choice travels with `LSHRS.save_to_disk` / pickle.  Every shape is modelled on both builds (bands of
                  two rows or more over fewer than 9 elements: the SkylakeX build's small-matrix kernels, model 3).
      (attributes, settable after construction)  ``spin_wait_us`` (2 000): how long a synchronous ``hash_device`` polls the pinned
                  `done` word of its launch before it sleeps on the stream (0: always the stream); ``audit_min_interval_s`` (0.05):
                  the live audit of ``audit_every`` runs at most this often, however short the batches are
      devices     in-process multi-device ingestion: host batches of >= 32 768 rows per device are cut into one row slice
                  per entry, hashed concurrently (one thread + hasher per entry), keys returned in row order
    """

    def __init__(self, num_bands: int, rows_per_band: int, dim: int, seed: int = 42, *, device=None,
                 tie_break: str = "host", tau_ulps=None, precision: str = "bf16x3",
                 tau1_ulps=None, tie_threads: Optional[int] = None,
                 tie_replay: str = "auto", margin_guard: float = 0.5, audit_every: int = 64,
                 audit_unflagged: int = 4096, devices: Optional[Sequence[int]] = None,
                 reference_blas: str = "host") -> None:
        # messages: lshrs/hash/lsh.py:78-83
        if num_bands <= 0:
            raise ValueError("num_bands must be > 0")
        if rows_per_band <= 0:
            raise ValueError("rows_per_band must be > 0")
        if dim <= 0:
            raise ValueError("dim must be > 0")
        if tie_break not in ("host", "none"):
            raise ValueError("tie_break must be 'host' or 'none'")
        if precision not in ("f32", "bf16x3"):
            raise ValueError("precision must be 'f32' or 'bf16x3'")
        self.num_bands = int(num_bands)
        self.rows_per_band = int(rows_per_band)
        self.dim = int(dim)
        self.tie_break = tie_break
        if reference_blas != "host":
            if reference_blas in _hostblas.LEGACY_BUILD_NAMES:
                raise ValueError(f"reference_blas={reference_blas!r} named the AVX2 (Zen 1 - 3) kernels only and is gone: say "
                                 f"{_hostblas.LEGACY_BUILD_NAMES[reference_blas]!r} for those, 'openblas-skylakex' for Zen 4 / "
                                 "Zen 5 (AVX-512); LSHHasher.host_blas_name() tells which one this host runs")
            if reference_blas not in _hostblas.NAMED_BUILDS:
                raise ValueError("reference_blas must be 'host' or one of " + ", ".join(sorted(_hostblas.NAMED_BUILDS)))
            if tie_replay != "auto":
                raise ValueError("a named reference_blas is replayed on the device: tie_replay must be 'auto'")
            if not _hostblas.named_model(reference_blas, self.rows_per_band, self.dim):
                raise ValueError(f"reference_blas={reference_blas!r} is not modelled for rows_per_band={self.rows_per_band}, "
                                 f"dim={self.dim}; use reference_blas='host'")
        self.reference_blas = reference_blas
        self._host_agrees: Optional[tuple] = None
        # In-process multi-device ingestion: host batches handed to `hash_batch_packed` are cut into one
        # contiguous row slice per entry of `devices`, each slice hashed by a worker thread through a hasher of its own
        # (own device, streams, scratch; the same hyperplanes), the keys land in one array in the original row order.  No
        # collective, nothing shared but the read-only hyperplanes.  An index may appear twice ([0, 0]: two slices in
        # flight on one device).  Everything else - device tensors, single vectors - runs on devices[0].
        self._devices = None if devices is None else [int(d) for d in devices]
        if self._devices is not None and (len(self._devices) == 0 or any(d < 0 for d in self._devices)):
            raise ValueError("devices must be a non-empty sequence of device indices")
        if self._devices is not None and device is None:
            device = self._devices[0]
        self._ctor_kwargs = dict(tie_break=tie_break, tau_ulps=tau_ulps, precision=precision, tau1_ulps=tau1_ulps,
                                 tie_threads=tie_threads, tie_replay=tie_replay, margin_guard=margin_guard,
                                 audit_every=audit_every, audit_unflagged=audit_unflagged, reference_blas=reference_blas)
        self._seed = seed
        self._children: Optional[list] = None
        self._pool = None
        self.multi_device_min_rows = 32_768      # per slice: below this a batch stays on devices[0]
        for name, value in (("tau_ulps", tau_ulps), ("tau1_ulps", tau1_ulps)):
            if isinstance(value, str) and value != "bound":
                raise ValueError(f"{name} must be a number, None or 'bound'")
        # "bound" (the default, also what None means): the PROVEN windows - per-hyperplane coefficients derived by
        # `window_coefficients`, handed to the library once per (hyperplanes, BLAS model): nothing is decided in stage 1
        # that could come out differently on the host.  A number: a window of that many units of 2^-24 ||x|| ||p||, a
        # statistical statement about the data ("measured"; the stage-1 one is watched by the margin guard).
        self.window_mode = {"tau": "bound" if tau_ulps in (None, "bound") else "measured",
                            "tau1": "bound" if tau1_ulps in (None, "bound") else "measured"}
        # (in "bound" mode these two hold the TYPICAL size of the proven window in the same units - for list capacities
        #  and reports; the kernels get LSHRS_WINDOW_PROVEN and use the coefficients.  Refined once the hyperplanes exist.)
        self.tau_ulps = bound_tau_ulps(self.dim) if tau_ulps in (None, "bound") else float(tau_ulps)
        # "bf16x3": large batches take the split-precision first pass (bf16 matrix cores, >2x the rate) followed by
        # the exact decision for every projection inside the stage-1 window; same keys as "f32" 
        self.precision = precision
        self.tau1_ulps = bound_tau1_ulps(self.dim) if tau1_ulps in (None, "bound") else float(tau1_ulps)
        self.window_info: Dict[str, object] = {}
        self._window_set: Dict[int, tuple] = {}
        self._window_coef_cache: Dict[tuple, tuple] = {}
        self.margin_guard = float(margin_guard)
        # every audit_every-th synchronous batch (and the first): a few of the projections the device has decided are
        # re-evaluated with NumPy on the host and compared (0 = never)
        self.audit_every = int(audit_every)
        # every launch of the split pass: this many of the projections stage 1 did NOT flag are replayed by stage 2 as well
        # and compared with what stage 1 decided (include/lshrs_hip.h, lshrs_sig_audit); 0 = off
        if int(audit_unflagged) < 0:
            raise ValueError("audit_unflagged must be >= 0")
        self.audit_unflagged = int(audit_unflagged)
        self._audit_seed = 0
        self.audit_totals = {"audited": 0, "sign_disagreements": 0, "max_window_ratio": 0.0}
        self._audit_countdown = 1
        self.audit_failures = 0
        self.margin_escalations = 0        # batches whose measured stage-1 deviation tripped the guard (then: bound window)
        # the split pass (two launches; 256-row workgroups, 128-row ones for short vectors and small batches) overtakes the f32 kernel at about 16 M input elements:
        # 20 k rows at 768-d, 8 k at 1536-d, 120 k at 128-d (tools/split_crossover.py)
        self.split_min_rows = 4_096
        self.split_min_elems = 16 << 20
        self.replay_min_rows = 256      # with the ties broken on the device the split pass pays from here (see _split_applies)
        # a synchronous `hash_device` polls the pinned `done` word of its launch for at most this long before it sleeps on the stream
        # (lshrs_wait_done): the runtime's wake-up is 10 - 20 us late - 10 % of a short-vector step; 0 = always the stream
        self.spin_wait_us = 2_000
        # host tie-break workers (lshrs_amd/_hostblas.py): None = this process's share of the cores (at most 8),
        # 1 = NumPy's batched matmul on the calling thread.  Same BLAS call either way.
        if tie_threads is not None and int(tie_threads) < 1:
            raise ValueError("tie_threads must be >= 1")
        self.tie_threads = None if tie_threads is None else int(tie_threads)
        self._host_planes_cache: Optional[Tuple[int, np.ndarray]] = None
        self._split_range_ok: Optional[Tuple[int, bool]] = None
        self._split_shape_ok: Optional[Tuple[int, bool]] = None
        self._device = device
        self._lock = threading.Lock()
        self._stream_lock = threading.Lock()     # one streamed host batch at a time (shared staging buffers); see hash_batch_packed
        self._one_lock = threading.Lock()
        self._one_queue: list = []
        self._one_leader = False
        self._projection_version = 0
        self._workspaces: Dict[int, Tuple[int, object]] = {}
        self.last_stats: Dict[str, int] = {}
        # set to a list to collect (start_event, end_event, rows) around every signature-kernel launch
        # (bench.py uses it to time the kernel on the stream it runs on)
        self.kernel_events: Optional[list] = None
        # device batches of >= 2 chunks take the pipelined path; 262144 rows = four full-chip rounds of the f32
        # kernel (128-row workgroups, two per CU) and of the split pass (256-row workgroups, one per CU, at these sizes): each
        # chunk boundary costs a kernel ramp-down/ramp-up, each chunk a fixed ~60 us of host work
        self.pipeline_chunk_rows = 262_144
        self.pipeline_pair_head = True     # long batches: full-size chunks at the head are launched two at a time
        # "auto": batches that take the split pass resolve their ties ON THE DEVICE, by replaying the host BLAS's
        # summation order in stage 2 - provided that order has been recognised on this host (the model is checked
        # bit for bit against `P_band @ x` of this process, _hostblas.blas_order_model) - and need no chunking, no
        # export and no host step.  "off": always the host engine / NumPy.  Same bytes either way.
        if tie_replay not in ("auto", "off"):
            raise ValueError("tie_replay must be 'auto' or 'off'")
        self.tie_replay = tie_replay
        self._replay_model_cache: Optional[tuple] = None
        self._route_memo: Optional[tuple] = None
        # the live audit against `P_band @ x` (audit_every): not more often than this, however short the batches are (it costs
        # ~0.3 ms - a small device gather, one copy, sixteen host sgemv calls: 3 % of a stream of 0.15 ms steps at every 64th)
        self.audit_min_interval_s = 0.05
        self._audit_last = 0.0
        # Stage 2 column by column (ABI 6, lshrs_sig_sort): every eight entries stage 2 takes share one hyperplane, fetched once -
        # the row gather of x is the only stream left.  "auto" = "buckets" for every hasher the split pass's main kernel serves
        # (up to 1024 padded key columns, rows longer than four k-tiles): stage 1 itself appends flagged and sampled projections
        # to the segment of their key column, no launch between the stages (config 2: stage 2 0.107 -> 0.084 ms; config 5: 3.36 ->
        # 2.37 ms).  False: the plain stage 2.  Same keys every way.  (Round 5's two other experiments - a device counting sort of
        # the plain list, and the pass cut into chunks whose stage 2 ran beside the next chunk's stage 1 - were slower than this
        # on every shape and plan measured (profiles/r05_chunk_overlap.log) and were removed in round 6.)
        self.stage2_sorted = "auto"
        self._sort_res: Dict[tuple, tuple] = {}
        self._bucket_cap_hint = 0
        self._pipes: Dict[tuple, int] = {}
        self._plan_cache: Dict[tuple, tuple] = {}
        self._replay_scratch: Dict[object, tuple] = {}
        self._async_pending: list = []
        self._replay_events: Dict[int, list] = {}
        self._pinned_cache: Dict[tuple, tuple] = {}
        self._small_epoch = 0
        self._flag_cap_hint = 0
        # hyperplanes: one generator, num_bands float64 draws cast to float32 (lsh.py:93-94)
        gen = np.random.default_rng(seed)
        planes = [gen.standard_normal((self.rows_per_band, self.dim)).astype(np.float32)
                  for _ in range(self.num_bands)]
        self._projections = _ProjectionList(planes, self)

    # ------------------------------------------------------------------ hyperplanes
    @property
    def projections(self) -> List[np.ndarray]:
        """Host-owned hyperplanes, one (rows_per_band, dim) float32 matrix per band.

        Source of truth: re-assigning the attribute (``load_from_disk`` and ``__setstate__`` of the
        orchestrator do, lshrs/core/main.py:981,1044) or an item of the list re-uploads the device copy on
        the next call.  After editing a matrix *in place* call :meth:`refresh_device`.
        """
        return self._projections

    @projections.setter
    def projections(self, value) -> None:
        self._projections = _ProjectionList(list(value), self)
        self._projection_version += 1

    def refresh_device(self) -> None:
        self._projection_version += 1

    @property
    def band_bytes(self) -> int:
        return (self.rows_per_band + 7) // 8

    def _stacked(self) -> np.ndarray:
        planes = [np.ascontiguousarray(p, dtype=np.float32) for p in self._projections]
        if len(planes) != self.num_bands or any(p.shape != (self.rows_per_band, self.dim) for p in planes):
            raise ValueError(
                f"projections must be {self.num_bands} arrays of shape ({self.rows_per_band}, {self.dim})")
        return np.concatenate(planes, axis=0)

    # ------------------------------------------------------------------ device plumbing
    def _torch_device(self, like=None):
        torch = _native.require_gpu()
        if like is not None:
            return like.device
        if self._device is None:
            return torch.device("cuda", torch.cuda.current_device())
        dev = torch.device(self._device) if not isinstance(self._device, int) else torch.device("cuda", self._device)
        if dev.index is None or dev.index >= torch.cuda.device_count():
            # (no index, or one this process does not have - an index unpickled on a worker with fewer GPUs or pinned to
            #  another one: the current device, as a hasher built without `device` uses)
            dev = torch.device("cuda", torch.cuda.current_device())
        return dev

    def _workspace(self, dev):
        """Device image of the hyperplanes in MFMA-fragment order (rebuilt when they change)."""
        torch = _native.require_gpu()
        lib = _native.load()
        cached = self._workspaces.get(dev.index)
        if cached is not None and cached[0] == self._projection_version:
            return cached[1]
        nbytes = lib.lshrs_sig_workspace_bytes(self.num_bands, self.rows_per_band, self.dim)
        if nbytes < 0:
            _native.check(int(nbytes), "lshrs_sig_workspace_bytes")
        with torch.cuda.device(dev):
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            p_dev = torch.from_numpy(self._stacked()).to(dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _native.check(
                lib.lshrs_sig_pack_projections(p_dev.data_ptr(), self.num_bands, self.rows_per_band, self.dim,
                                               ws.data_ptr(), stream),
                "lshrs_sig_pack_projections")
            torch.cuda.current_stream(dev).synchronize()  # p_dev may be freed after this
        self._workspaces[dev.index] = (self._projection_version, ws)
        self._window_set.pop(dev.index, None)     # (a fresh workspace: its window block sends everything to the exact decision)
        return ws

    def _tau_arg(self) -> float:
        """The tie window as the library takes it: LSHRS_WINDOW_PROVEN (0) or tau_ulps * 2^-24."""
        return 0.0 if self.window_mode["tau"] == "bound" else float(self.tau_ulps * _U)

    def _tau1_arg(self) -> float:
        return 0.0 if self.window_mode["tau1"] == "bound" else float(self.tau1_ulps * _U)

    def _ensure_window(self, dev, ws, model: int) -> None:
        """Hand the proven windows' coefficients for these hyperplanes and this decision engine (BLAS order model, 0 = f32
        chain + host engine) to the library, once per (device, hyperplanes, model)."""
        if "bound" not in self.window_mode.values():
            return
        key = (self._projection_version, int(model))
        if self._window_set.get(dev.index) == key:
            return
        torch = _native.require_gpu()
        lib = _native.load()
        cached = self._window_coef_cache.get(key)         # (a hasher that alternates between engines derives each once)
        if cached is None:
            ca, cb, ct, info = window_coefficients(self._stacked(), int(model), self.rows_per_band)
            cached = (np.stack([ca, cb, ct]), info)
            self._window_coef_cache = {k: v for k, v in self._window_coef_cache.items() if k[0] == key[0]}
            self._window_coef_cache[key] = cached
        coef_host, info = cached
        with torch.cuda.device(dev):
            if dev.index in self._window_set:
                torch.cuda.synchronize(dev)       # (a pass that reads the previous coefficients may still be running)
            coef = torch.from_numpy(coef_host).to(dev)
            stream = torch.cuda.current_stream(dev)
            _native.check(lib.lshrs_sig_set_window(ws.data_ptr(), self.num_bands, self.rows_per_band, self.dim,
                                                   coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(),
                                                   stream.cuda_stream), "lshrs_sig_set_window")
            stream.synchronize()                  # coef may be freed after this
        self._window_set[dev.index] = key
        self.window_info = info
        if self.window_mode["tau1"] == "bound":
            self.tau1_ulps = float(info["window_units"])
        if self.window_mode["tau"] == "bound":
            self.tau_ulps = float(info["tie_units"])

    # ------------------------------------------------------------------ core: device -> device
    def hash_device(self, x, *, out=None, row_flags=None, tie_break: Optional[str] = None, stats: Optional[dict] = None):
        """Hash a device-resident ``(n, dim)`` float32 ``torch.Tensor``.

        Returns a ``(n, num_bands, band_bytes)`` uint8 tensor on the same device.  ``row_flags``
        (optional uint8 tensor of n) receives bit0 = zero vector, bit1 = NaN present.  ``stats`` (optional dict) receives THIS
        call's statistics - ``last_stats`` is the hasher's last batch, which under concurrent callers may be another thread's
        (``hash_device_async(...).stats`` for the streaming form).
        """
        torch = _native.require_gpu()
        if x.dim() != 2 or x.shape[1] != self.dim:
            raise ValueError(f"Expected vectors of dimension {self.dim}, received {tuple(x.shape)}")
        if x.dtype != torch.float32 or not x.is_cuda:
            raise TypeError("hash_device expects a float32 CUDA/ROCm tensor")
        if x.stride(1) != 1:
            x = x.contiguous()
        mode = self.tie_break if tie_break is None else tie_break
        with self._lock:
            keys = self._hash_device_locked(x, out, row_flags, mode, host_rows=None, yield_lock=True)
            if stats is not None:
                stats.update(self.last_stats)
            return keys

    def _hash_device_locked(self, x, out, row_flags, mode, host_rows, allow_pipeline: bool = True, yield_lock: bool = False):
        torch = _native.require_gpu()
        lib = _native.load()
        dev = x.device
        n = int(x.shape[0])
        bb = self.band_bytes
        if out is None:
            out = torch.empty((n, self.num_bands, bb), dtype=torch.uint8, device=dev)
        elif out.shape != (n, self.num_bands, bb) or out.dtype != torch.uint8 or not out.is_contiguous():
            raise ValueError("out must be a contiguous uint8 tensor of shape (n, num_bands, band_bytes)")
        stats = {"n": n, "tie_entries": 0, "tie_pairs": 0, "relaunches": 0}
        self.last_stats = stats
        if n == 0:
            return out
        ws = self._workspace(dev)
        # (the route of a batch depends on a handful of facts that rarely change from call to call: remembered per set of them -
        #  the table of `_route` is consulted again when the hyperplanes, the BLAS's configuration or an option changes)
        ldx = x.stride(0)
        facts = (n if n < 131_072 else -1, mode, x.data_ptr() % 16 == 0 and ldx % 4 == 0, ldx < (1 << 20), host_rows is not None,
                 allow_pipeline, self._projection_version, self.tie_replay, self.precision, self.replay_min_rows, self.reference_blas,
                 _hostblas.blas_signature() if self.reference_blas == "host" else None)
        memo = self._route_memo
        if memo is not None and memo[0] == facts:
            route, model = memo[1]
        else:
            route, model = self._route(n, mode, aligned=facts[2], short_stride=facts[3], host_rows=host_rows is not None,
                                       allow_pipeline=allow_pipeline)
            if model:        # (the replay routes depend on nothing else; the host-engine routes also look at windows and engines)
                self._route_memo = (facts, (route, model))
        if route != "raw":                     # (the raw bits consult no window)
            self._ensure_window(dev, ws, model)
        tau = self._tau_arg()
        stats["route"] = route
        if route == "split+replay":
            return self._hash_device_replay(x, out, row_flags, ws, tau, stats, model, yield_lock)
        if route == "f32+replay":
            return self._hash_device_f32_replay(x, out, row_flags, ws, tau, stats, model)
        if route == "host-engine pipelined":
            return self._hash_device_pipelined(x, out, row_flags, ws, tau, stats)
        # "plain": one pass (split or f32 kernel), then the host decides the ties (engine or NumPy); "raw": no tie-break
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            flags_ptr = row_flags.data_ptr() if row_flags is not None else None
            if mode == "none":
                while True:
                    flag = self._launch_sig(torch, lib, dev, x.data_ptr(), n, x.stride(0), ws.data_ptr(),
                                            self.num_bands, self.rows_per_band, self.dim, out.data_ptr(), None, 0,
                                            None, 0.0, flags_ptr, stream)
                    if not self._flag_overflow(flag):
                        return out
                    stats["relaunches"] += 1
            cap = min(max(4096, n // 16 + 4096, int(1.3 * self._expected_tie_entries(n)) + 4096), 2 ** 30)
            while True:
                tie_list = torch.empty((cap, 2), dtype=torch.int64, device=dev)
                tie_count = torch.zeros(1, dtype=torch.int32, device=dev)
                flag = self._launch_sig(torch, lib, dev, x.data_ptr(), n, x.stride(0), ws.data_ptr(), self.num_bands,
                                        self.rows_per_band, self.dim, out.data_ptr(), tie_list.data_ptr(), cap,
                                        tie_count.data_ptr(), tau, flags_ptr, stream)
                cnt = int(tie_count.item())  # synchronises the stream
                overflow = self._flag_overflow(flag)
                if cnt <= cap and not overflow:
                    break
                cap = max(cap, cnt)  # the kernels counted every entry they wanted to write: relaunch with room
                stats["relaunches"] += 1
            stats["tie_entries"] = cnt
            if cnt and host_rows is None and self._tie_engine() is not None:
                # rows on the device, native engine: pairs and unique rows are cut on the device, the rows cross PCIe in
                # chunks through two pinned blocks while the engine works on the chunk before
                self._plain_resolve_device(torch, lib, dev, x, out, tie_list[:cnt], stats, stream)
            elif cnt:
                if cnt > 8192:                   # (long lists: the pairs are cut on the device - the same pairs in the same order)
                    rows_d, bands_d = self._tie_pairs_device(torch, tie_list[:cnt])
                    rows, bands = rows_d.cpu().numpy(), bands_d.to(torch.int32).cpu().numpy()
                else:
                    rows, bands = self._tie_pairs(tie_list[:cnt].cpu().numpy())
                stats["tie_pairs"] = int(rows.shape[0])
                urows, inverse = np.unique(rows, return_inverse=True)
                if host_rows is not None:
                    xh = host_rows(urows)
                else:
                    idx_dev = torch.from_numpy(urows).to(dev)
                    xg = torch.empty((urows.shape[0], self.dim), dtype=torch.float32, device=dev)
                    _native.check(
                        lib.lshrs_gather_rows_f32(x.data_ptr(), x.stride(0), self.dim, idx_dev.data_ptr(),
                                                  urows.shape[0], xg.data_ptr(), stream),
                        "lshrs_gather_rows_f32")
                    xh = xg.cpu().numpy()
                patch = self._tie_patches(xh, inverse, bands)
                rows_dev = torch.from_numpy(rows).to(dev)
                bands_dev = torch.from_numpy(bands).to(dev)
                patch_dev = torch.from_numpy(patch).to(dev)
                _native.check(
                    lib.lshrs_scatter_band_keys_u8(out.data_ptr(), self.num_bands, bb, rows_dev.data_ptr(),
                                                   bands_dev.data_ptr(), patch_dev.data_ptr(), rows.shape[0], stream),
                    "lshrs_scatter_band_keys_u8")
                torch.cuda.current_stream(dev).synchronize()  # the small staging tensors die with this frame
        return out

    # ------------------------------------------------------------------ which route a device batch takes
    _PLAIN_CHUNK_BYTES = 100 << 20        # per pinned block of the plain route's device -> host staging (two blocks; at most 32 768 rows each)

    def _tie_pairs_device(self, torch, entries):
        """`_tie_pairs` on the device: kernel tie entries ``(row*65536 + word, 32-bit column mask)`` (int64 pairs) -> the unique
        (row, band) pairs sorted by (band, row), as int64 tensors on the entries' device."""
        rows = entries[:, 0] >> 16
        words = entries[:, 0] & 0xFFFF
        band_cols = 8 * self.band_bytes
        codes = []
        if band_cols % 32 == 0:
            band = (32 * words) // band_cols
            keep = band < self.num_bands
            codes.append(band[keep] * (1 << 48) + rows[keep])
        else:
            masks = entries[:, 1]
            for c in range(32):
                hit = ((masks >> c) & 1).bool()
                band = (32 * words[hit] + c) // band_cols
                keep = band < self.num_bands
                codes.append(band[keep] * (1 << 48) + rows[hit][keep])
        code = torch.unique(torch.cat(codes))
        return code & ((1 << 48) - 1), code >> 48

    def _plain_resolve_device(self, torch, lib, dev, x, out, entries, stats, stream) -> None:
        """The plain route's tie-break for device-resident rows: what `_tie_pairs` + `np.unique` + one pageable copy of every tied
        row + one engine call did in sequence (0.2 s per 1 M x 768 rows at the proven windows), as a short pipeline."""
        eng, planes = self._tie_engine()
        rows_d, bands_d = self._tie_pairs_device(torch, entries)
        urows_d, inverse_d = torch.unique(rows_d, return_inverse=True)
        order = torch.argsort(inverse_d, stable=True)                   # pairs in the order of their rows
        rows_d, bands_d, inverse_d = rows_d[order], bands_d[order], inverse_d[order]
        m, u = int(rows_d.shape[0]), int(urows_d.shape[0])
        stats["tie_pairs"] = m
        inverse = inverse_d.to(torch.int32).cpu().numpy()
        bands = bands_d.to(torch.int32).cpu().numpy()
        chunk = max(1024, min(32_768, self._PLAIN_CHUNK_BYTES // (4 * self.dim)))
        key = ("plain", dev.index, self.dim)
        bufs = self._pinned_cache.get(key)
        if bufs is None:
            bufs = tuple(torch.empty((chunk, self.dim), dtype=torch.float32).pin_memory() for _ in range(2))
            self._pinned_cache[key] = bufs
        stage = torch.empty((2, chunk, self.dim), dtype=torch.float32, device=dev)
        events = [torch.cuda.Event(), torch.cuda.Event()]
        patch = np.empty((m, self.band_bytes), dtype=np.uint8)
        starts = list(range(0, u, chunk))
        bounds = np.searchsorted(inverse, np.asarray(starts + [u], dtype=np.int64), side="left")

        def send(i):            # chunk i: gather its rows on the device, start their copy into pinned block i % 2
            lo = starts[i]
            k = min(chunk, u - lo)
            _native.check(lib.lshrs_gather_rows_f32(x.data_ptr(), x.stride(0), self.dim, urows_d[lo:lo + k].data_ptr(), k,
                                                    stage[i % 2].data_ptr(), stream), "lshrs_gather_rows_f32")
            bufs[i % 2][:k].copy_(stage[i % 2][:k], non_blocking=True)
            events[i % 2].record(torch.cuda.current_stream(dev))

        if starts:
            send(0)
        for i, lo in enumerate(starts):
            if i + 1 < len(starts):
                send(i + 1)                                          # (block (i + 1) % 2 was read by the engine in step i - 1)
            events[i % 2].synchronize()
            plo, phi = int(bounds[i]), int(bounds[i + 1])
            k = min(chunk, u - lo)
            if phi > plo:
                patch[plo:phi] = eng.patch(planes, bufs[i % 2][:k].numpy(), inverse[plo:phi] - lo, bands[plo:phi])
        patch_dev = torch.from_numpy(patch).to(dev)
        bands32 = bands_d.to(torch.int32)
        _native.check(lib.lshrs_scatter_band_keys_u8(out.data_ptr(), self.num_bands, self.band_bytes, rows_d.data_ptr(),
                                                     bands32.data_ptr(), patch_dev.data_ptr(), m, stream),
                      "lshrs_scatter_band_keys_u8")
        torch.cuda.current_stream(dev).synchronize()    # the staging tensors die with this frame

    ROUTES = (
        # name                     taken when (first match wins)
        ("raw",                    "tie_break='none': the kernel's own bits, no tie-break"),
        ("split+replay",           "host BLAS order recognised, >= replay_min_rows rows, shape takes the split pass (>= 256 key columns "
                                   "or 128 .. 224 with dim >= 384 - or at most 256 key columns at dim <= 128 .. 256: the resident-image "
                                   "kernel -, hyperplane norms in range); rows of any length >= 9 (8 m + 4 elements: up to 4096) at any "
                                   "4-byte address; bands of one row included (round 6: stage 2 replays the host's sdot)"),
        ("f32+replay",             "host BLAS order recognised: small batches and shapes the split pass does not take - any dim "
                                   "(dim % 4 elements through the library's scalar tail), rows at any 4-byte address"),
        ("host-engine pipelined",  "no recognised BLAS order (or tie_replay='off'), >= 131 072 rows, the host engine exists, a tie window "
                                   "narrow enough for the per-chunk lists (measured windows): chunks overlapped by csrc/pipeline.hip, "
                                   "ties by the library's own sgemv"),
        ("plain",                  "everything else: one pass, then the host engine or NumPy on the tied pairs"),
    )

    def _route(self, n: int, mode: str, *, aligned: bool, short_stride: bool, host_rows: bool, allow_pipeline: bool = True):
        """(route name, BLAS order model the windows are set for) of a device batch - the one place that decides; the table
        above says why, tests/test_abi_and_boundary.py::test_route_table has a row per route."""
        if mode != "host":
            return "raw", 0
        model = self._replay_model() if self.tie_replay == "auto" else 0
        if model:       # (the model's own limits - fewer than 9 elements only on the Haswell / Zen build - are in `model`)
            # (round 5: rows at any 4-byte address - an offset view, 102 or 767 elements a row - through both stage-1 kernels)
            # (model 3 - the SkylakeX build's small-matrix kernels, at most eight elements - is replayed by the plain-load form only)
            if model != 3 and short_stride and self._split_applies(n, replay=True):
                return "split+replay", model
            # (the replay kernels follow whatever `model` licenses: every length at any 4-byte address - fewer than 9 elements only
            #  where the host's / the named build is the Haswell / Zen one, whose order is modelled down to one element)
            return "f32+replay", model
        if (self.reference_blas == "host" and allow_pipeline and not host_rows and n >= max(131_072, self.pipeline_chunk_rows // 2)
                and self._expected_tie_entries(32) <= 0.75 and self._tie_engine() is not None):
            # (the pipeline's per-chunk lists - and the pinned copies of the tied rows behind them - hold one entry per 32
            #  rows; the PROVEN tie window without a replay, ~1 000 units at 768-d, ties a third of the rows: every chunk
            #  would overflow and be hashed twice, so that case takes the plain path with a list sized for it)
            return "host-engine pipelined", 0
        if self.reference_blas != "host":      # (cannot happen for shapes the constructor accepted: the named order has no host engine)
            raise _native.NativeLibraryError(f"no device route replays reference_blas={self.reference_blas!r} for this batch")
        return "plain", 0

    def _expected_tie_entries(self, rows: int) -> float:
        """Projections a batch of Gaussian-like rows puts inside the tie window in force: per unit of window width
        (2^-24 ||x|| ||p||) a projection lands inside with probability 2 * 2^-24 * sqrt(dim / 2 pi) - 1.32e-6 at 768-d."""
        per_unit = 2.0 * _U * (self.dim / (2.0 * np.pi)) ** 0.5
        return rows * self.num_bands * self.rows_per_band * min(float(self.tau_ulps), 1.0e5) * per_unit

    # ------------------------------------------------------------------ ties broken on the device
    def _replay_model(self) -> int:
        """Summation-order model stage 2 replays for this hasher's shape: the host BLAS's, recognised and verified (0: not
        recognised -> host engine) - or, with a named `reference_blas`, that build's, whatever the host has."""
        if self.reference_blas != "host":
            return _hostblas.named_model(self.reference_blas, self.rows_per_band, self.dim)
        cached = self._replay_model_cache
        sig = _hostblas.blas_signature()      # (library, thread count, pid): one C call - the licence is per signature
        if cached is None or cached[0] != self._projection_version or cached[2] != sig:
            planes = self._stacked().reshape(self.num_bands, self.rows_per_band, self.dim)
            cached = (self._projection_version, int(_hostblas.blas_order_model(planes)), sig)
            self._replay_model_cache = cached
            if cached[1] == 0 and self.tie_replay != "off" and not getattr(self, "_warned_unrecognised", False):
                self._warned_unrecognised = True        # (once per hasher: the drop from the device route to the host engine is 40 x)
                import warnings

                warnings.warn("lshrs_amd: the summation order of this process's BLAS ("
                              f"{sig[0] or 'not found'}) is not one the "
                              "device replay knows: near-ties are decided by the host engine (same keys as the reference on this "
                              "host, ~40 x slower than the device route); reference_blas=\"openblas-skylakex\" / \"openblas-haswell\" "
                              "pins the keys to a named build and keeps the device route", HostBlasNotRecognised, stacklevel=3)
        return cached[1]

    @staticmethod
    def host_blas_name() -> Optional[str]:
        """Which named build (``reference_blas=`` value) this process's NumPy computes like - "openblas-skylakex" (AVX-512:
        Intel servers, AMD Zen 4 / 5) or "openblas-haswell" (AVX2: Haswell .., Zen 1 - 3) -, or None when its BLAS is not
        recognised.  What `LSHRS.save_to_disk` / pickle record beside ``reference_blas="host"``, so that loading the index on
        a host of the other kind can say so (``_hostblas.host_build_name``)."""
        return _hostblas.host_build_name()

    def _host_blas_agrees(self) -> bool:
        """Is what this process's NumPy computes the order the keys are pinned to?  (Always, for reference_blas="host"; for a
        named build: where the host's own BLAS is recognised as the same model.)  The live audit against `P_band @ x` only
        means something where it is."""
        if self.reference_blas == "host":
            return True
        sig = _hostblas.blas_signature()
        cached = self._host_agrees
        if cached is None or cached[0] != self._projection_version or cached[2] != sig:
            planes = self._stacked().reshape(self.num_bands, self.rows_per_band, self.dim)
            cached = (self._projection_version, int(_hostblas.blas_order_model(planes)) == self._replay_model(), sig)
            self._host_agrees = cached
        return cached[1]


    def _split_applies(self, n: int, replay: bool = False) -> bool:
        """Does a batch of n rows take the split-precision pass?  ``replay``: asked on behalf of the path that also
        breaks the ties on the device - it has no host step to amortise, and beats "f32 kernel + host tie-break" from
        a few hundred rows up (85 against 300 us at 512 x 768, tools/replay_crossover.py), so only tiny batches (a
        query vector: the fine-geometry f32 kernel answers in 35 us) stay off it."""
        if self.precision != "bf16x3" or (self.dim < 32 and not (replay and self._resident_shape())):
            return False
        if self.dim % 4 != 0 and not (replay and self.dim >= 9):
            return False              # (a scalar tail: stage 1 shifts it into place, the plain-load replay follows it - from 9 elements)
        if self.rows_per_band == 1 and not replay:      # (the host sums a one-row band with sdot: stage 2 of the REPLAYING pass follows that -
            return False                                #  round 6; the host-engine routes keep the f32 kernel for such bands)
        body = self.dim & ~3
        if body % 8 != 0 and body > 4096:  # (8 m + 4 elements beyond 4096: the library's short last block - the plain-load replay)
            return False
        if self.dim % 32 != 0 and not replay:      # (a partial last k-tile: only the replaying stage 2 masks the row's end)
            return False
        if replay:
            if n < self.replay_min_rows:
                return False
        elif n < self.split_min_rows or n * self.dim < self.split_min_elems:
            return False
        cached = self._split_shape_ok
        if cached is not None and cached[0] == self._projection_version:
            return cached[1]
        ok = self._split_shape_check()
        self._split_shape_ok = (self._projection_version, ok)
        return ok

    def _resident_shape(self) -> bool:
        """Short vectors of a narrow hasher - at most 256 key columns, dim <= 256, (16-column tiles) x (32-element k-tiles)
        <= 64: BASELINE config 1's 16 x 4 x 128, the reference's docstring layout 20 x 6 x 128, num_perm = 128 at 256-d -: stage
        1 of the split pass runs with the whole fragment image resident in LDS (sig16r_kernel; the library's `sig_resident`
        decides the same way)."""
        real = self.num_bands * self.rows_per_band
        if real > 256 or self.dim > 256 or self.dim < 8:
            return False
        nct = ((real + 15) // 16 + 3) // 4 * 4
        kt = 2 if self.dim <= 64 else (4 if self.dim <= 128 else 8)
        return nct * kt <= 64

    def _split_shape_check(self) -> bool:
        key_cols = 8 * self.num_bands * self.band_bytes
        resident = self._resident_shape()
        # >= 256 key columns - or 128 .. 224 (the reference's default num_perm = 128 as 8 x 16, config 1's 16 x 4, its
        # docstring's 20 x 6): those run on a fragment image zero-padded to 256 columns - up to half the matrix work wasted,
        # still 1.5x the f32 kernel
        if key_cols < 128 and not resident:
            return False
        if self.dim > 8192:
            # the proven window widens the row norms stage 1 accumulates in f32 by 0.1 %: enough for the rounding of up to
            # ~8 k terms (4 k dot2 steps x 2^-23); longer rows keep the f32 kernel
            return False
        if key_cols < 225 and self.dim < 384 and not resident:
            # short vectors: the padded pass's 256-column epilogue outweighs its matrix rate (1M x 128, 16 x 4:
            # 0.36 ms against 0.32 ms for the f32 kernel; 1M x 768, 8 x 16: 1.18 against 1.71 ms)
            return False
        # hyperplanes far outside the unit scale (user-assigned matrices) leave the range in which the bf16 split
        # and the window arithmetic are safe from under/overflow: those hashers keep the f32 kernel
        cached = self._split_range_ok
        if cached is None or cached[0] != self._projection_version:
            norms = np.sqrt((self._stacked().astype(np.float64) ** 2).sum(axis=1))
            live = norms[norms > 0]
            ok = bool(np.isfinite(norms).all() and (live.size == 0 or (live.min() >= 2.0 ** -40 and live.max() <= 2.0 ** 40)))
            cached = (self._projection_version, ok)
            self._split_range_ok = cached
        return cached[1]

    def project_device(self, x):
        """Diagnostic: raw f32 projections ``(n, num_perm)`` as the kernel accumulates them."""
        torch = _native.require_gpu()
        lib = _native.load()
        if x.dim() != 2 or x.shape[1] != self.dim or x.dtype != torch.float32 or not x.is_cuda:
            raise ValueError("project_device expects a float32 device tensor of shape (n, dim)")
        if x.stride(1) != 1:
            x = x.contiguous()
        dev = x.device
        n = int(x.shape[0])
        ldy = int(lib.lshrs_sig_padded_columns(self.num_bands, self.rows_per_band))
        y = torch.empty((n, ldy), dtype=torch.float32, device=dev)
        with self._lock, torch.cuda.device(dev):
            ws = self._workspace(dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _native.check(
                lib.lshrs_sig_project_f32(x.data_ptr(), n, x.stride(0), ws.data_ptr(), self.num_bands,
                                          self.rows_per_band, self.dim, y.data_ptr(), ldy, stream),
                "lshrs_sig_project_f32")
        band_cols = 8 * self.band_bytes
        cols = (torch.arange(self.num_bands, device=dev).repeat_interleave(self.rows_per_band) * band_cols
                + torch.arange(self.rows_per_band, device=dev).repeat(self.num_bands))
        return y[:, cols]

    # ------------------------------------------------------------------ validation (lsh.py:213-247)
    def _validate_vector(self, vector) -> np.ndarray:
        vec = np.asarray(vector, dtype=np.float32).reshape(-1)
        if vec.ndim != 1 or vec.shape[0] != self.dim:
            raise ValueError(f"Expected vector of dimension {self.dim}, received {vec.shape}")
        return vec

    # ------------------------------------------------------------------ pickling: host state only
    def close(self) -> None:
        """Release the native pipeline objects (device scratch, pinned host buffers, side stream)."""
        pool, self._pool = getattr(self, "_pool", None), None
        if pool is not None:
            pool.shutdown(wait=False)
        for child in getattr(self, "_children", None) or ():
            child.close()
        self._children = None
        self._pinned_cache = {}          # (the staging blocks of the host-fed and plain routes: pinned memory goes back with them)
        pipes, self._pipes = getattr(self, "_pipes", {}), {}
        if pipes:
            try:
                lib = _native.load()
                for pipe in pipes.values():
                    lib.lshrs_pipe_destroy(pipe)
            except Exception:  # pragma: no cover - interpreter shutdown
                pass

    def __del__(self):  # pragma: no cover - exercised implicitly
        try:
            self.close()
        except Exception:
            pass

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_lock"] = None
        state["_stream_lock"] = None
        state["_one_lock"] = None
        state["_one_queue"] = []
        state["_one_leader"] = False
        state["_workspaces"] = {}
        state["_children"] = None
        state["_pool"] = None
        state["_window_set"] = {}
        state["_window_coef_cache"] = {}
        state["_pinned_cache"] = {}
        state["_small_epoch"] = 0
        state["_pipes"] = {}
        state["_plan_cache"] = {}
        state["_replay_scratch"] = {}
        state["_sort_res"] = {}
        state["_async_pending"] = []
        state["_replay_events"] = {}
        state["_replay_model_cache"] = None
        state["_route_memo"] = None
        state["_host_agrees"] = None
        state["_host_planes_cache"] = None
        state["kernel_events"] = None
        state["_projections"] = list(self._projections)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.__dict__.setdefault("tie_threads", None)
        self.__dict__.setdefault("_pipes", {})
        self.__dict__.setdefault("_plan_cache", {})
        self.__dict__.setdefault("_replay_scratch", {})
        self.__dict__.setdefault("_sort_res", {})
        self.__dict__.setdefault("stage2_sorted", "auto")
        self.__dict__.setdefault("_bucket_cap_hint", 0)
        for gone in ("chunking", "chunk_min_rounds", "_chunk_res"):      # (options of round 5 a pickled hasher may still carry)
            self.__dict__.pop(gone, None)
        if self.__dict__.get("stage2_sorted") in ("sort", True):
            self.stage2_sorted = "auto"
        self.__dict__.setdefault("_async_pending", [])
        self.__dict__.setdefault("_replay_events", {})
        self.__dict__.setdefault("_replay_model_cache", None)
        self.__dict__.setdefault("_route_memo", None)
        self.__dict__.setdefault("audit_min_interval_s", 0.05)
        self.__dict__.setdefault("_audit_last", 0.0)
        self.__dict__.setdefault("tie_replay", "auto")
        self.__dict__.setdefault("reference_blas", "host")
        self.reference_blas = _hostblas.LEGACY_BUILD_NAMES.get(self.reference_blas, self.reference_blas)
        self.__dict__.setdefault("_host_agrees", None)
        self.__dict__.setdefault("replay_min_rows", 256)
        self.__dict__.setdefault("spin_wait_us", 2_000)
        self.__dict__.setdefault("pipeline_pair_head", True)
        self.__dict__.setdefault("_host_planes_cache", None)
        self.__dict__.setdefault("_split_range_ok", None)
        self.__dict__.setdefault("_split_shape_ok", None)
        self.__dict__.setdefault("split_min_elems", 16 << 20)
        self.__dict__.setdefault("margin_guard", 0.5)
        self.__dict__.setdefault("audit_every", 64)
        self.__dict__.setdefault("audit_unflagged", 4096)
        self.__dict__.setdefault("_audit_seed", 0)
        self.__dict__.setdefault("audit_totals", {"audited": 0, "sign_disagreements": 0, "max_window_ratio": 0.0})
        self.__dict__.setdefault("_small_epoch", 0)
        self.__dict__.setdefault("_audit_countdown", 1)
        self.__dict__.setdefault("audit_failures", 0)
        self.__dict__.setdefault("margin_escalations", 0)
        self.__dict__.setdefault("window_mode", {"tau": "measured", "tau1": "measured"})
        self.__dict__.setdefault("window_info", {})
        self.__dict__.setdefault("_devices", None)
        self.__dict__.setdefault("_children", None)
        self.__dict__.setdefault("_pool", None)
        self.__dict__.setdefault("_seed", 42)
        self.__dict__.setdefault("_ctor_kwargs", {})
        self.__dict__.setdefault("multi_device_min_rows", 32_768)
        self._window_set = {}
        self._window_coef_cache = {}
        self._lock = threading.Lock()
        self._stream_lock = threading.Lock()
        self._one_lock = threading.Lock()
        self._one_queue = []
        self._one_leader = False
        self._projections = _ProjectionList(state["_projections"], self)
