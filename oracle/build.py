"""Build + load the C part of the oracle (oracle/chain_model.c → oracle/_build/libchain_model.so).

TEST INFRASTRUCTURE ONLY.  gcc, no dependencies.  ``-ffp-contract=off`` so the
compiler neither fuses nor splits anything: every rounding in the model is an
explicit ``fmaf``.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "chain_model.c")
_OUT_DIR = os.path.join(_HERE, "_build")
_OUT = os.path.join(_OUT_DIR, "libchain_model.so")
_lock = threading.Lock()
_lib = None


def build(force: bool = False) -> str:
    """Compile if missing or stale; return the .so path."""
    with _lock:
        if not force and os.path.exists(_OUT) and os.path.getmtime(_OUT) >= os.path.getmtime(_SRC):
            return _OUT
        os.makedirs(_OUT_DIR, exist_ok=True)
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", _SRC, "-o", _OUT + ".tmp", "-lm"]
        subprocess.run(cmd, check=True)
        os.replace(_OUT + ".tmp", _OUT)
        return _OUT


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        path = build()
        lib = ctypes.CDLL(path)
        f32p = ctypes.POINTER(ctypes.c_float)
        lib.lshrs_chain_project.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p, ctypes.c_int, f32p]
        lib.lshrs_chain_project.restype = None
        lib.lshrs_chain_hash.argtypes = [
            f32p, ctypes.c_int64, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_int,
            ctypes.POINTER(ctypes.c_uint8),
        ]
        lib.lshrs_chain_hash.restype = None
        lib.lshrs_cosine_f64.argtypes = [
            f32p, f32p, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
        ]
        lib.lshrs_cosine_f64.restype = None
        _lib = lib
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _stack(projections) -> np.ndarray:
    return _f32(np.concatenate([np.asarray(p, dtype=np.float32) for p in projections], axis=0))


def chain_project(projections, vectors) -> np.ndarray:
    """(n, num_perm) f32 projections in the kernel's fmaf order."""
    lib = load()
    x = _f32(vectors)
    p = _stack(projections)
    y = np.empty((x.shape[0], p.shape[0]), dtype=np.float32)
    f32p = ctypes.POINTER(ctypes.c_float)
    lib.lshrs_chain_project(x.ctypes.data_as(f32p), x.shape[0], x.shape[1], p.ctypes.data_as(f32p), p.shape[0],
                            y.ctypes.data_as(f32p))
    return y


def chain_hash_packed(projections, vectors) -> np.ndarray:
    """(n, num_bands, ceil(rows/8)) u8 raw kernel-order keys (no tie-break)."""
    lib = load()
    x = _f32(vectors)
    p = _stack(projections)
    nb = len(projections)
    r = projections[0].shape[0]
    out = np.empty((x.shape[0], nb, (r + 7) // 8), dtype=np.uint8)
    f32p = ctypes.POINTER(ctypes.c_float)
    lib.lshrs_chain_hash(x.ctypes.data_as(f32p), x.shape[0], x.shape[1], p.ctypes.data_as(f32p), nb, r,
                         out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    return out


def cosine_f64(query, candidates) -> np.ndarray:
    lib = load()
    q = _f32(query).reshape(-1)
    c = _f32(candidates)
    s = np.empty(c.shape[0], dtype=np.float64)
    f32p = ctypes.POINTER(ctypes.c_float)
    lib.lshrs_cosine_f64(q.ctypes.data_as(f32p), c.ctypes.data_as(f32p), c.shape[0], c.shape[1],
                         s.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return s


if __name__ == "__main__":
    print(build(force=True))
