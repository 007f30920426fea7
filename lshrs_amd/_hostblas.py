"""Locate the ``cblas_sgemv`` of the BLAS library NumPy itself uses, for the native tie-break loop.

The tie-break must reproduce ``projection @ vector`` (lshrs/hash/lsh.py:200) *as this host's NumPy
evaluates it*.  NumPy turns that expression into one ``cblas_sgemv`` call on its bundled OpenBLAS;
``csrc/host_tiebreak.cpp`` issues the same call through the same library, minus NumPy's per-call
dispatch.  Nothing here is trusted blindly: ``verified_for(rows, dim)`` compares the native loop
with NumPy bit for bit on random data of the actual shape, and the caller falls back to the NumPy
expression when the symbol is missing or a single bit differs.
"""

from __future__ import annotations

import ctypes
import os
import re
import threading
from typing import Dict, Optional, Tuple

import numpy as np

_lock = threading.Lock()
_resolved: Optional[Tuple[int, int, object, Optional[int]]] = None
_searched = False
_verified: Dict[Tuple[int, int], bool] = {}

_SYMBOLS = (("scipy_cblas_sgemv64_", 1), ("cblas_sgemv64_", 1), ("scipy_cblas_sgemv", 0), ("cblas_sgemv", 0))


def sgemv_pointer() -> Optional[Tuple[int, int, Optional[int]]]:
    """(address, ilp64, address of openblas_set_num_threads_local or None) of NumPy's BLAS, or None."""
    global _resolved, _searched
    with _lock:
        if not _searched:
            _searched = True
            np.ones((2, 2), dtype=np.float32) @ np.ones(2, dtype=np.float32)  # make sure BLAS is mapped
            paths = []
            try:
                with open("/proc/self/maps") as fh:
                    for line in fh:
                        m = re.search(r"(/\S*numpy\S*(?:openblas|blas)\S*\.so\S*)", line)
                        if m and m.group(1) not in paths:
                            paths.append(m.group(1))
            except OSError:
                paths = []
            for path in paths:
                try:
                    lib = ctypes.CDLL(path)
                except OSError:
                    continue
                for name, ilp64 in _SYMBOLS:
                    fn = getattr(lib, name, None)
                    if fn is not None:
                        local = None
                        for lname in ("openblas_set_num_threads_local", "scipy_openblas_set_num_threads_local64_",
                                      "scipy_openblas_set_num_threads_local"):
                            lfn = getattr(lib, lname, None)
                            if lfn is not None:
                                local = ctypes.cast(lfn, ctypes.c_void_p).value
                                break
                        _resolved = (ctypes.cast(fn, ctypes.c_void_p).value, ilp64, lib, local)
                        break
                if _resolved:
                    break
        return None if _resolved is None else (_resolved[0], _resolved[1], _resolved[3])


def band_keys(lib, planes, xrows: np.ndarray, xindex: np.ndarray, bands: np.ndarray, rows: int, dim: int,
              threads: int, want_y: bool = False):
    """Run the native loop.  ``planes``: list of C-contiguous float32 (rows, dim) arrays."""
    ptr = sgemv_pointer()
    if ptr is None:
        return None
    m = int(bands.shape[0])
    bb = (rows + 7) // 8
    patch = np.empty((m, bb), dtype=np.uint8)
    y = np.empty((m, rows), dtype=np.float32) if want_y else None
    arr = (ctypes.c_void_p * len(planes))(*[p.ctypes.data for p in planes])
    xrows = np.ascontiguousarray(xrows, dtype=np.float32)
    xindex = np.ascontiguousarray(xindex, dtype=np.int64)
    bands = np.ascontiguousarray(bands, dtype=np.int32)
    code = lib.lshrs_host_band_keys_f32(ptr[0], ptr[1], ptr[2], arr, rows, dim, xrows.ctypes.data, xindex.ctypes.data,
                                        bands.ctypes.data, m, patch.ctypes.data,
                                        y.ctypes.data if y is not None else None, threads)
    if code != 0:
        return None
    return (patch, y) if want_y else patch


def verified_for(lib, rows: int, dim: int) -> bool:
    """True once the native loop has reproduced NumPy's ``P @ x`` bit for bit for this shape."""
    key = (rows, dim)
    with _lock:
        if key in _verified:
            return _verified[key]
    ok = False
    if sgemv_pointer() is not None:
        rng = np.random.default_rng(rows * 100003 + dim)
        planes = [np.ascontiguousarray(rng.standard_normal((rows, dim)).astype(np.float32)) for _ in range(3)]
        xs = rng.standard_normal((48, dim)).astype(np.float32)
        xs[5] = 0.0
        xs[6, : max(1, dim // 2)] *= 1e-20
        bands = np.repeat(np.arange(3, dtype=np.int32), 48)
        xindex = np.tile(np.arange(48, dtype=np.int64), 3)
        bands = np.tile(bands, 8)      # enough pairs (1152) that the worker-thread path is the one verified
        xindex = np.tile(xindex, 8)
        got = band_keys(lib, planes, xs, xindex, bands, rows, dim, threads=4, want_y=True)
        if got is not None:
            patch, y = got
            want_y = np.stack([planes[b] @ xs[i] for b, i in zip(bands, xindex)])
            want_p = np.packbits(want_y > 0, axis=1, bitorder="little")
            ok = bool(np.array_equal(y.view(np.uint32), want_y.view(np.uint32)) and np.array_equal(patch, want_p))
    with _lock:
        _verified[key] = ok
    return ok


def worker_threads() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    return max(1, min(16, n))
