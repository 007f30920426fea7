"""ctypes binding of ``csrc/liblshrs_host.so`` (C ABI: ``include/lshrs_host.h``): the host tie-break engine.

The engine re-evaluates the (row, band) pairs the GPU pass flagged with the reference's own BLAS call
(``projection @ vector``, lshrs/hash/lsh.py:200) on several cores at once — every worker thread calls
``cblas_sgemv`` through a private mapping of the library NumPy itself is linked against, because callers that
share one OpenBLAS mapping are serialised by its buffer lock.  No arithmetic of its own: if the library NumPy
uses cannot be located, or a shape fails the self-check against ``P_band @ x``, the caller keeps using NumPy's
batched ``matmul`` (same call, one core).
"""

from __future__ import annotations

import ctypes
import os
import subprocess
import threading
from typing import Dict, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
SOURCE = os.path.join(_HERE, "csrc", "host_tiebreak.cpp")
LIBRARY = os.path.join(_HERE, "csrc", "liblshrs_host.so")
INCLUDE = os.path.join(REPO_ROOT, "include")
ABI_VERSION = 2
EXPORTS = ("lshrs_host_abi_version", "lshrs_tb_create", "lshrs_tb_threads", "lshrs_tb_destroy", "lshrs_tb_patch",
           "lshrs_tb_resolve", "lshrs_tb_model_dot", "lshrs_tb_model_row_dot")

# (cblas_sgemv symbol, 64-bit integers?, set_num_threads symbol) in order of preference
_BLAS_FLAVOURS = (
    ("scipy_cblas_sgemv64_", 1, "scipy_openblas_set_num_threads64_"),   # NumPy >= 2 wheels
    ("cblas_sgemv64_", 1, "openblas_set_num_threads64_"),               # NumPy 1.2x wheels
    ("cblas_sgemv", 0, "openblas_set_num_threads"),                     # system OpenBLAS
)

_lock = threading.Lock()
_lib: Optional[ctypes.CDLL] = None
_engine: Optional["TieBreakEngine"] = None
_engine_failed = False
_engine_pid = -1


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the host engine with g++ into the in-tree shared library."""
    with _lock:
        newest = max(os.path.getmtime(SOURCE), os.path.getmtime(os.path.join(INCLUDE, "lshrs_host.h")))
        if not force and os.path.exists(LIBRARY) and os.path.getmtime(LIBRARY) >= newest:
            return LIBRARY
        cxx = os.environ.get("CXX", "g++")
        cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + INCLUDE, SOURCE, "-o", LIBRARY + ".tmp",
               "-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        os.replace(LIBRARY + ".tmp", LIBRARY)
        return LIBRARY


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        lib = ctypes.CDLL(LIBRARY)
        for name in EXPORTS:
            if not hasattr(lib, name):
                raise OSError(f"{LIBRARY} does not export {name}; rebuild it")
        c = ctypes
        lib.lshrs_host_abi_version.restype = c.c_int
        lib.lshrs_tb_create.argtypes = [c.c_char_p, c.c_char_p, c.c_char_p, c.c_int, c.c_int]
        lib.lshrs_tb_create.restype = c.c_void_p
        lib.lshrs_tb_threads.argtypes = [c.c_void_p]
        lib.lshrs_tb_threads.restype = c.c_int
        lib.lshrs_tb_destroy.argtypes = [c.c_void_p]
        lib.lshrs_tb_destroy.restype = None
        lib.lshrs_tb_patch.argtypes = [c.c_void_p, c.c_void_p, c.c_int32, c.c_int32, c.c_int32, c.c_void_p, c.c_int64,
                                       c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p]
        lib.lshrs_tb_patch.restype = c.c_int
        lib.lshrs_tb_resolve.argtypes = [c.c_void_p, c.c_void_p, c.c_int32, c.c_int32, c.c_int32, c.c_void_p, c.c_int64,
                                         c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p]
        lib.lshrs_tb_resolve.restype = c.c_int
        lib.lshrs_tb_model_dot.argtypes = [c.c_void_p, c.c_void_p, c.c_int64, c.c_int32]
        lib.lshrs_tb_model_dot.restype = c.c_float
        lib.lshrs_tb_model_row_dot.argtypes = [c.c_void_p, c.c_void_p, c.c_int64, c.c_int32, c.c_int32, c.c_int32]
        lib.lshrs_tb_model_row_dot.restype = c.c_float
        if lib.lshrs_host_abi_version() != ABI_VERSION:
            raise OSError(f"{LIBRARY} has a different ABI version; rebuild it")
        _lib = lib
        return lib


def numpy_blas() -> Optional[Tuple[str, str, int, str]]:
    """(path, sgemv symbol, ilp64, set_num_threads symbol) of the BLAS this process's NumPy calls, or None."""
    np.dot(np.ones((2, 2), np.float32), np.ones(2, np.float32))  # make sure it is mapped
    paths = []
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                path = line.rsplit(None, 1)[-1]
                if "openblas" in os.path.basename(path).lower() and path not in paths and os.path.exists(path):
                    paths.append(path)
    except OSError:
        return None
    # prefer the copy that lives next to numpy
    paths.sort(key=lambda p: (0 if "numpy" in p else 1, p))
    for path in paths:
        try:
            handle = ctypes.CDLL(path)
        except OSError:
            continue
        for sym, ilp64, setter in _BLAS_FLAVOURS:
            if hasattr(handle, sym):
                return path, sym, ilp64, setter if hasattr(handle, setter) else ""
    return None


_order_models: Dict[tuple, int] = {}
_blas_probe: Optional[tuple] = None      # (path, getter function or None), found once per process


def blas_signature() -> tuple:
    """(library path, thread count) of the BLAS this process's NumPy calls, right now.  OpenBLAS picks its sgemv
    micro-kernels per thread slice, so the summation order of ``P_band @ x`` may change when the thread count does
    (threadpoolctl, ``openblas_set_num_threads``, a forked worker with another affinity): the order model is
    licensed per signature and re-verified when it changes.  Costs one C call (~1 us)."""
    global _blas_probe
    if _blas_probe is None:
        found = numpy_blas()
        getter = None
        if found is not None:
            try:
                handle = ctypes.CDLL(found[0])
                for sym in ("scipy_openblas_get_num_threads64_", "openblas_get_num_threads64_", "openblas_get_num_threads"):
                    if hasattr(handle, sym):
                        getter = getattr(handle, sym)
                        getter.restype = ctypes.c_int
                        getter.argtypes = []
                        break
            except OSError:
                getter = None
        _blas_probe = (found[0] if found is not None else "", getter)
    path, getter = _blas_probe
    return (path, int(getter()) if getter is not None else -1, os.getpid())


def blas_row_kinds(rows_per_band: int) -> np.ndarray:
    """Which of OpenBLAS's sgemv_t micro-kernels computes each row of an r-row band (``lshrs_tb_model_row_dot``): 0 = the
    8-lane fma kernel (rows in groups of four), 1 = the 4x2 kernel (a pair of left-over rows), 2 = the 4x1 kernel (a single
    left-over row, or the third of three)."""
    r = int(rows_per_band)
    r4 = r & ~3
    j = np.arange(r)
    return np.where(j < r4, 0, np.where(((r & 3) == 1) | (j - r4 == 2), 2, 1)).astype(np.int32)


# Named builds of the BLAS the reference's `projection @ vector` (lshrs/hash/lsh.py:200) may run on: `LSHHasher(reference_blas=...)`
# pins the keys to one of them whatever BLAS this host's NumPy has.  Value = the model id of `lshrs_tb_model_row_dot` that build
# follows where the two differ (the dim % 4 elements of sgemv_t's scalar tail, the SIMD kernel of sdot, fewer than 9 elements).
# Which CPUs run which (OpenBLAS 0.3.2x DYNAMIC_ARCH, what NumPy's wheels ship - `host_build_name()` answers for THIS process):
#   "openblas-skylakex"  every CPU with AVX-512: the SkylakeX / Cooperlake / SapphireRapids core types - Intel Skylake-X, Cascade
#                        Lake, Ice Lake, Sapphire Rapids and later servers, AND AMD Zen 4 / Zen 5 (EPYC 9004 / 9005, Ryzen 7000+:
#                        the GPU box's EPYC 9575F is licensed as this build);
#   "openblas-haswell"   AVX2 without AVX-512: the Haswell / Zen core types - Intel Haswell .. client parts, AMD Zen 1 - 3
#                        (EPYC 7001 - 7003, Ryzen 1000 - 5000).
# (Round 5 had an alias "openblas-zen" for the second: wrong for the Zen generations this package targets, hence gone - an index
#  that recorded it is read as "openblas-haswell", which is what it computed.)
NAMED_BUILDS = {
    "openblas-skylakex": 1,
    "openblas-haswell": 2,
}
LEGACY_BUILD_NAMES = {"openblas-zen": "openblas-haswell"}
_MODEL_NAMES = {1: "openblas-skylakex", 2: "openblas-haswell"}
_build_names: Dict[tuple, Optional[str]] = {}


def host_build_name() -> Optional[str]:
    """Which NAMED build this process's NumPy computes like, right now - "openblas-skylakex", "openblas-haswell" - or None
    (a BLAS whose order is not recognised, or the host library has not been built).  Decided where the two builds differ: a
    band of seven rows over 102 elements (sgemv_t's scalar tail) and a one-row band over 100 (sdot), each licensed bit for bit
    against `P_band @ x` by `blas_order_model`; both must name the same build.  Cached per (library, thread count, process)."""
    sig = blas_signature()
    if sig not in _build_names:
        rng = np.random.default_rng(20241005)
        tail = blas_order_model(rng.standard_normal((1, 7, 102)).astype(np.float32))
        sdot = blas_order_model(rng.standard_normal((1, 1, 100)).astype(np.float32))
        _build_names[sig] = _MODEL_NAMES.get(tail) if tail == sdot else None
    return _build_names[sig]


def builds_differ(rows_per_band: int, dim: int) -> bool:
    """Do the named builds sum a band of this shape differently (so that keys of tied projections can differ between an
    index built on one and a query hashed on the other)?  True for a scalar tail (dim % 4 != 0), one-row bands (sdot) and
    fewer than 9 elements; False for whole groups of four elements in bands of two rows or more - every BASELINE config."""
    return named_model("openblas-skylakex", rows_per_band, dim) != named_model("openblas-haswell", rows_per_band, dim)


def named_model(build: str, rows_per_band: int, dim: int) -> int:
    """The summation-order model `lshrs_tb_model_row_dot` takes for hyperplane bands of this shape when the keys are pinned to
    a NAMED build of OpenBLAS (0: not a named build).  Every shape is modelled on both builds: one-row bands (sdot) at every
    length; bands of two rows or more over whole groups of four elements sum alike on both (model 1); a scalar tail
    (dim % 4 != 0, from 9 elements) follows the build; fewer than 9 elements: the SkylakeX build's small-matrix kernels are
    model 3, the Haswell build runs its usual kernels down to one element.  The coverage is what `blas_order_model` licenses on
    a host that really runs that build - tests/test_reference_blas.py checks the two against each other under
    ``OPENBLAS_CORETYPE``."""
    b = NAMED_BUILDS.get(build, 0)
    r, dim = int(rows_per_band), int(dim)
    if not b or r < 1 or dim < 1:
        return 0
    if r == 1:                       # sdot: every length, the build's own SIMD kernel
        return b
    body = dim & ~3
    if dim < 9 and b != 2:           # (fewer than 9 elements: the SkylakeX build's small-matrix kernels - model 3; the Haswell / Zen
        return 3                     #  build runs the kernels it runs for longer rows, at every length from 1)
    return 1 if dim % 4 == 0 else b  # (whole groups of four: both builds sum alike)


def blas_order_model(planes: np.ndarray) -> int:
    """Which summation-order model of ``lshrs_tb_model_row_dot`` (0 = none) reproduces, bit for bit, what this process's
    NumPy returns for ``P_band @ x`` at this ``(rows_per_band, dim)`` - the licence for the GPU's tie replay
    (``lshrs_sig_hash_batch_split_replay_f32``) to stand in for the host engine.  Checked on random vectors, on vectors
    with a wide dynamic range and on vectors built to cancel against a hyperplane (where the order shows), for the
    first, a middle and the last band, every row of each (a band of 13 rows goes through three different kernels of the
    library); cached per (shape, BLAS library, BLAS thread count, process): identity of the keys means identity with THIS
    process's BLAS as it is configured when the batch is hashed."""
    nb, r, dim = planes.shape
    key = (r, dim) + blas_signature()
    if key in _order_models:
        return _order_models[key]
    model = 0
    try:
        body = dim & ~3               # (dim % 4 elements behind it: the library's scalar tail - from 9 elements up on its SkylakeX
        # build, at every length on its Haswell / Zen build; what the candidates below do not reproduce is not licensed)
        # (a band of ONE row is sdot on the host: modelled for every length, round 5)
        if r >= 1 and os.path.exists(LIBRARY):
            lib = load()
            rng = np.random.default_rng(20240601)
            bands = sorted({0, nb // 2, nb - 1})
            kinds = blas_row_kinds(r)
            # rows to cancel against: the first, a middle one, the last, and the first row of every kernel kind
            targets = sorted({0, r // 2, r - 1} | {int(np.argmax(kinds == k)) for k in set(kinds.tolist())})
            trials = []
            for b in bands:
                plane = np.ascontiguousarray(planes[b], dtype=np.float32)
                xs = [rng.standard_normal(dim) for _ in range(12)]
                xs += [rng.standard_normal(dim) * np.exp(3.0 * rng.standard_normal(dim)) for _ in range(6)]
                for i in targets:                              # nearly orthogonal to row i: y_i is what the order leaves of it
                    p = plane[i].astype(np.float64)
                    if not (p @ p) > 0:
                        continue
                    for _ in range(3):
                        x = rng.standard_normal(dim)
                        xs.append(x - (x @ p) / (p @ p) * p)
                for x in xs:
                    x32 = np.ascontiguousarray(x, dtype=np.float32)
                    trials.append((plane, x32, plane @ x32))    # the reference's call (lshrs/hash/lsh.py:200)
            # model 2 differs from 1 only in the dim % 4 elements of the scalar tail (the library's Haswell / Zen build
            # contracts nothing there): tried second, and only where there is a tail
            # ... and in how a band of ONE row is summed (NumPy calls sdot there: the SIMD kernel of the build, then the
            # elements behind the last whole 32 in a double - lshrs_tb_model_row_dot)
            # ... model 3: the SkylakeX build's small-matrix kernels (bands of two rows and more over at most eight elements)
            for candidate in ((1, 2, 3) if dim < 9 and r >= 2 else ((1,) if dim % 4 == 0 and r >= 2 else (1, 2))):
                if all(np.array_equal(want.view(np.uint32), np.array(
                        [lib.lshrs_tb_model_row_dot(plane[i].ctypes.data, x32.ctypes.data, dim, candidate, i, r)
                         for i in range(r)], dtype=np.float32).view(np.uint32)) for plane, x32, want in trials):
                    model = candidate
                    break
    except OSError:
        model = 0
    _order_models[key] = model
    return model


def _core_budget() -> int:
    """Cores this process may actually burn: the affinity mask, capped by the cgroup CPU quota (a container on a
    256-core host often sees all 256 in its mask but is throttled beyond 16)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        cores = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            fields = open(path).read().split()
            if path.endswith("cpu.max"):
                if fields[0] != "max":
                    cores = min(cores, max(1, int(fields[0]) // int(fields[1])))
            else:
                quota = int(fields[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    cores = min(cores, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return cores


def default_threads() -> int:
    """Worker count: half of this process's share of the core budget (the ranks of one node divide it; the workers
    poll between chunks, and the Python thread, the HIP runtime and NumPy need cores too), at most 8."""
    forced = os.environ.get("LSHRS_TIE_THREADS")
    if forced:
        return max(1, min(64, int(forced)))
    ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    return max(1, min(8, _core_budget() // ranks // 2))


class TieBreakEngine:
    """Thread pool over private BLAS copies; ``patch`` is the drop-in for the NumPy tie-break loop."""

    def __init__(self, threads: int) -> None:
        info = numpy_blas()
        if info is None:
            raise OSError("the BLAS library NumPy calls could not be located (not an OpenBLAS build?)")
        self.blas_path, self.symbol, self.ilp64, setter = info
        self._lib = load()
        self._handle = self._lib.lshrs_tb_create(self.blas_path.encode(), self.symbol.encode(), setter.encode(),
                                                 self.ilp64, int(threads))
        if not self._handle:
            raise OSError(f"could not map a private copy of {self.blas_path}")
        self.threads = int(self._lib.lshrs_tb_threads(self._handle))
        # for the native pipeline driver (lshrs_pipe_hash_f32 calls lshrs_tb_resolve through this pointer)
        self.resolve_fn = ctypes.cast(self._lib.lshrs_tb_resolve, ctypes.c_void_p).value
        self._shape_ok: Dict[Tuple[int, int], bool] = {}

    @property
    def handle(self) -> int:
        return self._handle

    def close(self) -> None:
        if self._handle:
            self._lib.lshrs_tb_destroy(self._handle)
            self._handle = None

    def patch(self, planes: np.ndarray, xrows: np.ndarray, row_index: np.ndarray, bands: np.ndarray,
              want_y: bool = False):
        """planes (num_bands, r, dim) f32 C-contiguous; xrows (m, dim) f32 with contiguous rows;
        row_index/bands int32 (pairs).  Returns the (pairs, band_bytes) uint8 keys [and the projections]."""
        nb, r, dim = planes.shape
        m = int(bands.shape[0])
        keys = np.empty((m, (r + 7) // 8), dtype=np.uint8)
        y = np.empty((m, r), dtype=np.float32) if want_y else None
        if m:
            assert planes.dtype == np.float32 and planes.flags.c_contiguous
            assert xrows.dtype == np.float32 and xrows.strides[1] == 4 and xrows.shape[1] == dim
            row_index = np.ascontiguousarray(row_index, dtype=np.int32)
            bands = np.ascontiguousarray(bands, dtype=np.int32)
            if row_index.size and int(row_index.max()) >= xrows.shape[0]:
                raise IndexError("tie-break row index out of range")
            rc = self._lib.lshrs_tb_patch(self._handle, planes.ctypes.data, nb, r, dim, xrows.ctypes.data,
                                          xrows.strides[0] // 4, row_index.ctypes.data, bands.ctypes.data, m,
                                          keys.ctypes.data, y.ctypes.data if want_y else None)
            if rc != 0:
                raise ValueError("lshrs_tb_patch: bad argument")
        return (keys, y) if want_y else keys

    def resolve(self, planes: np.ndarray, entries_ptr: int, n_entries: int, xstage_ptr: int, ldx: int,
                rows_ptr: int, bands_ptr: int, keys_ptr: int, out_cap: int) -> int:
        """One pipeline chunk in one native call (raw pointers into pinned staging buffers): tie entries + staged
        vectors in, unique (row, band) pairs and their patched band keys out.  Returns the number of pairs."""
        nb, r, dim = planes.shape
        n_pairs = ctypes.c_int64(0)
        rc = self._lib.lshrs_tb_resolve(self._handle, planes.ctypes.data, nb, r, dim, entries_ptr, int(n_entries),
                                        xstage_ptr, int(ldx), rows_ptr, bands_ptr, keys_ptr, int(out_cap),
                                        ctypes.byref(n_pairs))
        if rc != 0:
            raise ValueError(f"lshrs_tb_resolve: bad argument or {n_pairs.value} pairs exceed the room for {out_cap}")
        return int(n_pairs.value)

    def shape_trusted(self, planes: np.ndarray) -> bool:
        """Self-check, once per (rows_per_band, dim): the engine's projections must equal ``P_band @ x`` of this
        process's NumPy bit for bit (guards against a BLAS whose result depends on its thread count)."""
        nb, r, dim = planes.shape
        key = (r, dim)
        if key not in self._shape_ok:
            rng = np.random.default_rng(20240229)
            xs = rng.standard_normal((48, dim)).astype(np.float32)
            bands = (np.arange(48) % nb).astype(np.int32)
            rows = np.arange(48, dtype=np.int32)
            _, y = self.patch(planes, xs, rows, bands, want_y=True)
            ok = all(np.array_equal((planes[b] @ xs[i]).view(np.uint32), y[i].view(np.uint32))
                     for i, b in enumerate(bands))
            self._shape_ok[key] = ok
        return self._shape_ok[key]


def engine(threads: Optional[int] = None) -> Optional[TieBreakEngine]:
    """The process-wide engine, or None when it cannot be built here (the caller then uses NumPy's matmul)."""
    global _engine, _engine_failed, _engine_pid
    if _engine is not None and _engine_pid != os.getpid():
        # forked child: the worker threads did not come along (the handle of the parent's engine is abandoned)
        _engine = None
        _engine_failed = False
    if _engine is not None or _engine_failed:
        return _engine
    want = default_threads() if threads is None else int(threads)
    if want < 2:
        _engine_failed = True   # one core: NumPy's own call is the same thing
        return None
    try:
        if not os.path.exists(LIBRARY):
            raise OSError(f"{LIBRARY} has not been built")
        eng = TieBreakEngine(want)
    except OSError:
        _engine_failed = True
        return None
    with _lock:
        if _engine is None:
            _engine = eng
            _engine_pid = os.getpid()
    return _engine
