"""ctypes binding of ``csrc/liblshrs_hip.so`` (the C ABI declared in ``include/lshrs_hip.h``).

There is no CPU implementation behind this module: if the shared library has not
been built, or the ABI version does not match, loading raises — loudly — and so
does every compute entry point of the package.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
import threading
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
# one translation unit per kernel family (what they share: csrc/lshrs_common.h); pipeline.hip: the native driver of the host-engine route
UNITS = ("sig_setup", "sig_f32", "sig16", "sig16r", "sig_replay", "sig_small", "sig_split", "storage", "rerank", "query", "pipeline")
SOURCES = tuple(os.path.join(CSRC, u + ".hip") for u in UNITS)
SOURCE = SOURCES[0]
# (LSHRS_HIP_LIBRARY: load another build of the same ABI instead - A/B measurements of compiler flags, tools/ab_build.py)
LIBRARY = os.environ.get("LSHRS_HIP_LIBRARY") or os.path.join(CSRC, "liblshrs_hip.so")
INCLUDE = os.path.join(REPO_ROOT, "include")
ABI_VERSION = 7
QUERY_MAX_PAIRS = 16384   # LSHRS_QUERY_MAX_PAIRS
SIG_COUNTERS = 8          # LSHRS_SIG_COUNTERS of include/lshrs_hip.h
SIG_DEVICE_COUNTERS = SIG_COUNTERS + 6 * 4096      # LSHRS_SIG_DEVICE_COUNTERS: the device block (counters + stage-2 slots)
SMALL_MAX_ROWS = 256      # LSHRS_SMALL_MAX_ROWS
SORT_MAX_COLS = 1024      # kSortMaxCols of csrc/lshrs_common.h: padded key columns the column-wise stage 2 takes

BUILD_WRONG_KEYS = 0x1   # LSHRS_BUILD_WRONG_KEYS
BUILD_TUNED = 0x2        # LSHRS_BUILD_TUNED

E_BADARG = -10001
E_TOOLARGE = -10002

_lock = threading.Lock()
_lib: Optional[ctypes.CDLL] = None


class NativeLibraryError(RuntimeError):
    """The HIP extension is missing, stale or reported an error."""


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into the in-tree shared library (and the host tie-break engine).
    One object per translation unit (kernels + C ABI, native pipeline driver), rebuilt only when stale."""
    from . import _hostblas

    _hostblas.build(force=force, verbose=verbose)
    with _lock:
        header = max(os.path.getmtime(os.path.join(INCLUDE, "lshrs_hip.h")), os.path.getmtime(os.path.join(CSRC, "lshrs_common.h")))
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        objdir = os.path.join(CSRC, "_obj")
        os.makedirs(objdir, exist_ok=True)
        objects, stale, relink = [], [], force or not os.path.exists(LIBRARY)
        for src in SOURCES:
            obj = os.path.join(objdir, os.path.basename(src) + ".o")
            objects.append(obj)
            if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), header):
                stale.append((src, obj))

        def compile_one(job):
            src, obj = job
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility-inlines-hidden", "-I" + INCLUDE,
                   "-I" + CSRC, "-c", src, "-o", obj + ".tmp"]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
            os.replace(obj + ".tmp", obj)

        if stale:       # (the units are independent: a few at a time - the two stage-1 kernels take ~7 s each, the rest 1-2 s)
            from concurrent.futures import ThreadPoolExecutor

            with ThreadPoolExecutor(max_workers=min(4, len(stale))) as pool:
                list(pool.map(compile_one, stale))
            relink = True
        if relink or os.path.getmtime(LIBRARY) < max(os.path.getmtime(o) for o in objects):
            cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *objects, "-o", LIBRARY + ".tmp"]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
            os.replace(LIBRARY + ".tmp", LIBRARY)
        return LIBRARY


def _declare(lib: ctypes.CDLL) -> None:
    c = ctypes
    vp, i32, i64, f32 = c.c_void_p, c.c_int32, c.c_int64, c.c_float
    lib.lshrs_abi_version.argtypes = []
    lib.lshrs_abi_version.restype = c.c_int
    lib.lshrs_build_flags.argtypes = []
    lib.lshrs_build_flags.restype = c.c_uint32
    lib.lshrs_sig_workspace_bytes.argtypes = [i32, i32, i32]
    lib.lshrs_sig_workspace_bytes.restype = i64
    lib.lshrs_sig_padded_columns.argtypes = [i32, i32]
    lib.lshrs_sig_padded_columns.restype = i32
    lib.lshrs_sig_pack_projections.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.lshrs_sig_pack_projections.restype = c.c_int
    # (workspace, bands, rows, dim, coef_a, coef_b, coef_tie, stream)
    lib.lshrs_sig_set_window.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
    lib.lshrs_sig_set_window.restype = c.c_int
    # (X, n, ldx, workspace, bands, rows, dim, keys, tie_list, tie_cap, tie_count, tau, row_flags, opts, stream)
    lib.lshrs_sig_hash_batch_f32.argtypes = [vp, i64, i64, vp, i32, i32, i32, vp, vp, i32, vp, f32, vp, vp, vp]
    lib.lshrs_sig_hash_batch_f32.restype = c.c_int
    # (... row_flags, flag_list, flag_cap, flag_count, tau1, opts, stream)
    lib.lshrs_sig_hash_batch_split_f32.argtypes = [vp, i64, i64, vp, i32, i32, i32, vp, vp, i32, vp, f32, vp, vp, i32, vp,
                                                   f32, vp, vp]
    lib.lshrs_sig_hash_batch_split_f32.restype = c.c_int
    # (X, n, ldx, workspace, bands, rows, dim, keys, counters, tau, row_flags, flag_list, flag_y, flag_cap, tau1,
    #  blas_model, host_counts, audit, opts, stream)
    lib.lshrs_sig_hash_batch_split_replay_f32.argtypes = [vp, i64, i64, vp, i32, i32, i32, vp, vp, f32, vp, vp, vp, i32, f32,
                                                          i32, vp, vp, vp, vp]
    lib.lshrs_sig_hash_batch_split_replay_f32.restype = c.c_int
    # (X, n, ldx, workspace, bands, rows, dim, keys, tie_list, tie_cap, counters, tau, flag_list, flag_cap, blas_model,
    #  host_counts, stream)
    lib.lshrs_sig_resolve_ties_replay_f32.argtypes = [vp, i64, i64, vp, i32, i32, i32, vp, vp, i32, vp, f32, vp, i32, i32, vp,
                                                      vp]
    lib.lshrs_sig_resolve_ties_replay_f32.restype = c.c_int
    # (X, n, ldx, workspace, bands, rows, dim, keys, row_flags, counters, tau, blas_model, host_done, epoch, stream)
    lib.lshrs_sig_hash_small_replay_f32.argtypes = [vp, i64, i64, vp, i32, i32, i32, vp, vp, vp, f32, i32, vp, i32, vp]
    lib.lshrs_sig_hash_small_replay_f32.restype = c.c_int
    lib.lshrs_stream_synchronize.argtypes = [vp]
    lib.lshrs_stream_synchronize.restype = c.c_int
    lib.lshrs_wait_done.argtypes = [vp, i32, i64, vp]
    lib.lshrs_wait_done.restype = c.c_int
    lib.lshrs_sig_project_f32.argtypes = [vp, i64, i64, vp, i32, i32, i32, vp, i64, vp]
    lib.lshrs_sig_project_f32.restype = c.c_int
    lib.lshrs_gather_rows_f32.argtypes = [vp, i64, i32, vp, i64, vp, vp]
    lib.lshrs_gather_rows_f32.restype = c.c_int
    lib.lshrs_gather_tied_rows_f32.argtypes = [vp, i64, i32, vp, vp, i32, vp, vp]
    lib.lshrs_gather_tied_rows_f32.restype = c.c_int
    lib.lshrs_scatter_band_keys_u8.argtypes = [vp, i32, i32, vp, vp, vp, i64, vp]
    lib.lshrs_scatter_band_keys_u8.restype = c.c_int
    lib.lshrs_keys_to_hex_u8.argtypes = [vp, i64, vp, vp]
    lib.lshrs_keys_to_hex_u8.restype = c.c_int
    lib.lshrs_copy_to_host_u8.argtypes = [vp, vp, i64, vp]
    lib.lshrs_copy_to_host_u8.restype = c.c_int
    lib.lshrs_bucket_histogram_u8.argtypes = [vp, i64, i32, i32, vp, vp]
    lib.lshrs_bucket_histogram_u8.restype = c.c_int
    lib.lshrs_bucket_scatter_u8.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp, vp]
    lib.lshrs_bucket_scatter_u8.restype = c.c_int
    lib.lshrs_cosine_batch_f32.argtypes = [vp, i64, i64, i32, vp, i32, vp, i32, vp, vp, vp, vp]
    lib.lshrs_cosine_batch_f32.restype = c.c_int
    lib.lshrs_l2_normalize_f32.argtypes = [vp, i64, i64, i32, vp, vp, vp]
    lib.lshrs_l2_normalize_f32.restype = c.c_int
    lib.lshrs_topk_workspace_bytes.argtypes = [i32, i32]
    lib.lshrs_topk_workspace_bytes.restype = i64
    lib.lshrs_topk_desc_f32.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
    lib.lshrs_topk_desc_f32.restype = c.c_int
    f64 = c.c_double
    # (keys, q, bands, band_bytes, segments, nseg, slot_start, slot_len, slot_off, pair_count, stream)
    lib.lshrs_query_lookup_u8.argtypes = [vp, i32, i32, i32, vp, i32, vp, vp, vp, vp, vp]
    lib.lshrs_query_lookup_u8.restype = c.c_int
    # (counts, q, top_k, top_p, keep_out, offsets, totals, stream)
    lib.lshrs_query_scan_i32.argtypes = [vp, i32, i32, f64, vp, vp, vp, vp]
    lib.lshrs_query_scan_i32.restype = c.c_int
    # (segments, nseg, bands, slot_start, slot_len, slot_off, pair_off, q, max_pairs, cand_ids, cand_hits, ucount, stream)
    lib.lshrs_query_collide_index_i64.argtypes = [vp, i32, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]
    lib.lshrs_query_collide_index_i64.restype = c.c_int
    # (members, bands, pair_off, q, max_pairs, num_bands, cand_ids, cand_hits, ucount, stream)
    lib.lshrs_query_collide_pairs_i64.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    lib.lshrs_query_collide_pairs_i64.restype = c.c_int
    lib.lshrs_query_big_workspace_bytes.argtypes = [i64]
    lib.lshrs_query_big_workspace_bytes.restype = i64
    # (segments, nseg, bands, slot_start, slot_off, pairs, workspace, cand_ids, cand_hits, ucount, stream)
    lib.lshrs_query_collide_big_i64.argtypes = [vp, i32, i32, vp, vp, i64, vp, vp, vp, vp, vp]
    lib.lshrs_query_collide_big_i64.restype = c.c_int
    # (keys, bands, band_bytes, segments, nseg, slot_start, slot_len, slot_off, max_pairs, top_k, top_p, rerank_follows, pair_off,
    #  cand_ids, ucount, keep, out_off, out_ids, done_host, epoch, copy_src, copy_dst, copy_n, stream)
    lib.lshrs_query_one_u8.argtypes = [vp, i32, i32, vp, i32, vp, vp, vp, i32, i32, f64, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp]
    lib.lshrs_query_one_u8.restype = c.c_int
    # (corpus, m, ldc, dim, queries, q, cand_rows, row_off, row_cnt, total, scores, err, stream)
    lib.lshrs_cosine_ragged_f32.argtypes = [vp, i64, i64, i32, vp, i32, vp, vp, vp, i64, vp, vp, vp]
    lib.lshrs_cosine_ragged_f32.restype = c.c_int
    # (cand_ids, scores, pair_off, ucount, keep, out_off, q, max_candidates, out_ids, out_scores, done_host, epoch, stream)
    lib.lshrs_query_rank_f32.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp]
    lib.lshrs_query_rank_f32.restype = c.c_int
    lib.lshrs_pipe_create.argtypes = [i32, i32, i32, i32, i32]
    lib.lshrs_pipe_create.restype = vp
    lib.lshrs_pipe_destroy.argtypes = [vp]
    lib.lshrs_pipe_destroy.restype = None
    lib.lshrs_pipe_hash_f32.argtypes = [vp, vp, i64, vp, vp, vp, f32, f32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.lshrs_pipe_hash_f32.restype = c.c_int


EXPORTS = (
    "lshrs_abi_version",
    "lshrs_build_flags",
    "lshrs_sig_workspace_bytes",
    "lshrs_sig_padded_columns",
    "lshrs_sig_pack_projections",
    "lshrs_sig_set_window",
    "lshrs_sig_hash_batch_f32",
    "lshrs_sig_hash_batch_split_f32",
    "lshrs_sig_hash_batch_split_replay_f32",
    "lshrs_sig_resolve_ties_replay_f32",
    "lshrs_sig_hash_small_replay_f32",
    "lshrs_stream_synchronize",
    "lshrs_wait_done",
    "lshrs_sig_project_f32",
    "lshrs_gather_rows_f32",
    "lshrs_gather_tied_rows_f32",
    "lshrs_scatter_band_keys_u8",
    "lshrs_keys_to_hex_u8",
    "lshrs_copy_to_host_u8",
    "lshrs_bucket_histogram_u8",
    "lshrs_bucket_scatter_u8",
    "lshrs_cosine_batch_f32",
    "lshrs_l2_normalize_f32",
    "lshrs_topk_workspace_bytes",
    "lshrs_topk_desc_f32",
    "lshrs_query_lookup_u8",
    "lshrs_query_scan_i32",
    "lshrs_query_collide_index_i64",
    "lshrs_query_collide_pairs_i64",
    "lshrs_query_big_workspace_bytes",
    "lshrs_query_collide_big_i64",
    "lshrs_query_one_u8",
    "lshrs_cosine_ragged_f32",
    "lshrs_query_rank_f32",
    "lshrs_pipe_create",
    "lshrs_pipe_destroy",
    "lshrs_pipe_hash_f32",
)


def load() -> ctypes.CDLL:
    """Load the library (once).  Raises NativeLibraryError when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIBRARY):
            raise NativeLibraryError(
                f"{LIBRARY} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
                "lshrs_amd has no CPU fallback."
            )
        # torch ships its own libamdhip64 (same SONAME); importing it first makes our
        # NEEDED entry resolve to the runtime torch allocates with.
        import torch  # noqa: F401

        try:
            lib = ctypes.CDLL(LIBRARY)
        except OSError as exc:  # pragma: no cover - environment specific
            raise NativeLibraryError(f"cannot load {LIBRARY}: {exc}") from exc
        for name in EXPORTS:
            if not hasattr(lib, name):
                raise NativeLibraryError(f"{LIBRARY} does not export {name}; rebuild it")
        _declare(lib)
        got = lib.lshrs_abi_version()
        if got != ABI_VERSION:
            raise NativeLibraryError(f"{LIBRARY} has ABI version {got}, expected {ABI_VERSION}; rebuild it")
        flags = int(lib.lshrs_build_flags())
        if flags & BUILD_WRONG_KEYS and os.environ.get("LSHRS_ALLOW_AB") != "1":
            # an A/B build that drops work the keys need (tools/ab_build.py -DLSHRS_AB_...): indistinguishable from the
            # product by ABI number, so it says what it is and is refused here
            raise NativeLibraryError(
                f"{LIBRARY} is a measurement build whose keys are wrong by design (lshrs_build_flags() = {flags:#x}); "
                "set LSHRS_ALLOW_AB=1 to load it for an A/B run, or unset LSHRS_HIP_LIBRARY")
        _lib = lib
        return lib


class SigOpts(ctypes.Structure):
    """``lshrs_sig_opts`` of include/lshrs_hip.h: optional per-call measurement hooks (events, clock probe)."""

    _fields_ = [("struct_bytes", ctypes.c_uint32), ("done_epoch", ctypes.c_int32),
                ("ev_stage1_start", ctypes.c_void_p), ("ev_stage1_stop", ctypes.c_void_p),
                ("ev_stage2_start", ctypes.c_void_p), ("ev_stage2_stop", ctypes.c_void_p),
                ("clock_probe", ctypes.c_void_p), ("sort", ctypes.c_void_p), ("done_host", ctypes.c_void_p)]

    def __init__(self, events=None, clock_probe=None, sort=None):
        super().__init__()
        self._sort_ref = None
        self.struct_bytes = ctypes.sizeof(SigOpts)
        if events is not None:
            (self.ev_stage1_start, self.ev_stage1_stop, self.ev_stage2_start, self.ev_stage2_stop) = events
        if clock_probe is not None:
            self.clock_probe = clock_probe
        self.set_sort(sort)

    def set_sort(self, sort) -> None:
        """Attach (or detach) the scratch of the column-sorted stage 2; the SigSort object must outlive the call it is passed to."""
        self._sort_ref = sort
        self.sort = ctypes.addressof(sort) if sort is not None else None


class SigSort(ctypes.Structure):
    """``lshrs_sig_sort`` of include/lshrs_hip.h: scratch for the column-sorted stage 2."""

    _fields_ = [("struct_bytes", ctypes.c_uint32), ("cap", ctypes.c_int32), ("list", ctypes.c_void_p), ("y", ctypes.c_void_p),
                ("hist", ctypes.c_void_p), ("thr", ctypes.c_void_p), ("mode", ctypes.c_int32), ("parity", ctypes.c_int32)]

    def __init__(self, list_ptr: int, y_ptr: int, hist_ptr: int, cap: int, thr_ptr=None, mode: int = 1):
        super().__init__()
        self.struct_bytes = ctypes.sizeof(SigSort)
        self.list, self.y, self.hist, self.cap = list_ptr, y_ptr, hist_ptr, int(cap)
        self.thr, self.mode, self.parity = thr_ptr, int(mode), 0


class SigAudit(ctypes.Structure):
    """``lshrs_sig_audit`` of include/lshrs_hip.h: where a launch leaves its sample of the projections stage 1 decided on
    its own, for stage 2 to verify against the replayed host-BLAS value."""

    _fields_ = [("struct_bytes", ctypes.c_uint32), ("seed", ctypes.c_uint32), ("list", ctypes.c_void_p),
                ("vals", ctypes.c_void_p), ("slots", ctypes.c_int32), ("target", ctypes.c_int32)]

    def __init__(self, list_ptr: int, vals_ptr: int, slots: int, target: int, seed: int = 0):
        super().__init__()
        self.struct_bytes = ctypes.sizeof(SigAudit)
        self.list, self.vals, self.slots, self.target, self.seed = list_ptr, vals_ptr, int(slots), int(target), int(seed) & 0xFFFFFFFF


def check(code: int, what: str) -> None:
    """Turn a C-ABI status into an exception."""
    if code == 0:
        return
    if code == E_BADARG:
        raise NativeLibraryError(f"{what}: bad argument (LSHRS_E_BADARG)")
    if code == E_TOOLARGE:
        raise NativeLibraryError(f"{what}: shape outside kernel limits (LSHRS_E_TOOLARGE)")
    raise NativeLibraryError(f"{what}: HIP error {-code}")


_torch_ok = None


def require_gpu():
    """Return the torch module after checking a GPU is usable; raise loudly otherwise."""
    global _torch_ok
    if _torch_ok is not None:          # (checked once per process: this sits on the per-call path of the hasher)
        return _torch_ok
    import torch

    if not torch.cuda.is_available():
        raise NativeLibraryError(
            "no MI355X/ROCm device is visible to PyTorch; lshrs_amd computes only on the GPU "
            "(there is no CPU fallback)."
        )
    _torch_ok = torch
    return torch
