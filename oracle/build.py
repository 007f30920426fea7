"""Build + load the C parts of the oracle (oracle/chain_model.c → oracle/_build/libchain_model.so, oracle/mfma_model.c →
oracle/_build/libmfma_model.so).

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


# ---- oracle/mfma_model.c: the arithmetic of v_mfma_f32_16x16x32_{bf16,f16} and of the split pass's accumulator -----------
_MFMA_SRC = os.path.join(_HERE, "mfma_model.c")
_MFMA_OUT = os.path.join(_OUT_DIR, "libmfma_model.so")
_mfma_lib = None


def build_mfma(force: bool = False) -> str:
    with _lock:
        if not force and os.path.exists(_MFMA_OUT) and os.path.getmtime(_MFMA_OUT) >= os.path.getmtime(_MFMA_SRC):
            return _MFMA_OUT
        os.makedirs(_OUT_DIR, exist_ok=True)
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", _MFMA_SRC, "-o",
               _MFMA_OUT + ".tmp", "-lm"]
        subprocess.run(cmd, check=True)
        os.replace(_MFMA_OUT + ".tmp", _MFMA_OUT)
        return _MFMA_OUT


def load_mfma() -> ctypes.CDLL:
    global _mfma_lib
    if _mfma_lib is None:
        lib = ctypes.CDLL(build_mfma())
        lib.lshrs_mfma16_model_batch.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int64]
        lib.lshrs_mfma16_model_batch.restype = None
        lib.lshrs_split_stage1_model.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        lib.lshrs_split_stage1_model.restype = ctypes.c_float
        lib.lshrs_split_stage1_model_batch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                                       ctypes.c_int, ctypes.c_void_p]
        lib.lshrs_split_stage1_model_batch.restype = None
        _mfma_lib = lib
    return _mfma_lib


def mfma16_model(kind: int, a_bits, b_bits, c) -> np.ndarray:
    """D of one v_mfma_f32_16x16x32 output element per case: kind 0 = f16, 1 = bf16; a_bits, b_bits (n, 32) uint16."""
    lib = load_mfma()
    A = np.ascontiguousarray(a_bits, dtype=np.uint16)
    B = np.ascontiguousarray(b_bits, dtype=np.uint16)
    C = _f32(c)
    D = np.zeros(C.shape[0], dtype=np.float32)
    lib.lshrs_mfma16_model_batch(int(kind), A.ctypes.data, B.ctypes.data, C.ctypes.data, D.ctypes.data, C.shape[0])
    return D


def split_stage1_model(projections, vectors) -> np.ndarray:
    """(n, num_perm) f32: the value stage 1 of the split pass (sig16_kernel) holds for every projection - the bf16x3
    split and the instruction model above, k-tile by k-tile, terms in the kernel's order."""
    lib = load_mfma()
    x = _f32(vectors)
    p = _stack(projections)
    if x.shape[1] % 32:                 # the kernel reads the chunks past a row's end as zero, the hyperplanes are zero-padded
        pad = 32 - x.shape[1] % 32
        x = np.ascontiguousarray(np.pad(x, ((0, 0), (0, pad))))
        p = np.ascontiguousarray(np.pad(p, ((0, 0), (0, pad))))
    y = np.empty((x.shape[0], p.shape[0]), dtype=np.float32)
    lib.lshrs_split_stage1_model_batch(x.ctypes.data, x.shape[0], p.ctypes.data, p.shape[0], x.shape[1], y.ctypes.data)
    return y


if __name__ == "__main__":
    print(build(force=True))
    print(build_mfma(force=True))
