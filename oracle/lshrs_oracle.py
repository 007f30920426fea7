"""NumPy restatement of the lshrs hot path — the parity ORACLE.

TEST INFRASTRUCTURE ONLY.  Nothing under ``lshrs_amd/`` imports this module; it
exists so that ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` have an independent CPU statement of what the reference
computes.  Each function names the reference lines it restates (paths are
relative to the upstream checkout, ``/root/reference`` in the build container).

Pinning: ``tests/golden/make_golden.py`` imported the *real* reference modules
in the build container and wrote ``tests/golden/*.npz|json``;
``tests/test_oracle_golden.py`` checks every function here against those
fixtures, so the oracle is pinned to the reference's own outputs ("parity
pinned by reference-generated goldens" — the reference's test-suite itself
holds no concrete signature bytes).

The arithmetic lives in NumPy/OpenBLAS exactly as it does for the reference
(``projection @ vector`` is a ``cblas_sgemv`` of shape (rows_per_band, dim)),
so on any one machine the *literal* functions below are bit-identical to the
reference by construction: same library, same call, same shapes.
"""

from __future__ import annotations

import math
from typing import Iterable, List, Sequence, Tuple

import numpy as np

__all__ = [
    "make_projections",
    "project_and_pack",
    "OracleHasher",
    "hash_vector_literal",
    "hash_batch_literal_packed",
    "hash_batch_blas_packed",
    "exact_projection_f64",
    "l2_norm",
    "cosine_similarity",
    "top_k_cosine",
    "prepare_vector",
    "is_zero_vector_rows",
    "band_bytes",
]


# --------------------------------------------------------------------------
# signature pass
# --------------------------------------------------------------------------

def band_bytes(rows_per_band: int) -> int:
    """Bytes per band key = ceil(rows_per_band / 8) (np.packbits zero-pads; lshrs/hash/lsh.py:208)."""
    return (int(rows_per_band) + 7) // 8


def make_projections(num_bands: int, rows_per_band: int, dim: int, seed: int = 42) -> List[np.ndarray]:
    """Hyperplanes exactly as lshrs/hash/lsh.py:78-94 draws them.

    One ``default_rng(seed)``; ``num_bands`` successive float64 ``standard_normal``
    draws of shape (rows_per_band, dim), each cast to float32.
    """
    if num_bands <= 0:
        raise ValueError("num_bands must be > 0")
    if rows_per_band <= 0:
        raise ValueError("rows_per_band must be > 0")
    if dim <= 0:
        raise ValueError("dim must be > 0")
    gen = np.random.default_rng(seed)
    out = []
    for _ in range(num_bands):
        out.append(gen.standard_normal((rows_per_band, dim)).astype(np.float32))
    return out


def project_and_pack(projection: np.ndarray, vector: np.ndarray) -> bytes:
    """One band key: lshrs/hash/lsh.py:200 (sgemv), :204 (strict > 0), :208 (LSB-first pack), :211."""
    y = projection @ vector
    bits = (y > 0).astype(np.uint8)
    return np.packbits(bits, bitorder="little").tobytes()


def _as_vector(vector, dim: int) -> np.ndarray:
    """lshrs/hash/lsh.py:241-247 — float32, flattened, length must equal dim."""
    v = np.asarray(vector, dtype=np.float32).reshape(-1)
    if v.ndim != 1 or v.shape[0] != dim:
        raise ValueError(f"Expected vector of dimension {dim}, received {v.shape}")
    return v


def hash_vector_literal(projections: Sequence[np.ndarray], vector, dim: int) -> Tuple[bytes, ...]:
    """lshrs/hash/lsh.py:129-134 — validate, then one project_and_pack per band in band order."""
    v = _as_vector(vector, dim)
    return tuple(project_and_pack(p, v) for p in projections)


def hash_batch_literal_packed(projections: Sequence[np.ndarray], vectors) -> np.ndarray:
    """Reference ``hash_batch`` (lshrs/hash/lsh.py:162-169) with the keys gathered into
    a uint8 array of shape (n, num_bands, ceil(rows/8)) instead of Python objects.

    Per vector and per band it issues the reference's exact NumPy calls, so the
    bytes equal ``LSHHasher.hash_batch`` byte for byte on the same machine.
    """
    arr = np.asarray(vectors, dtype=np.float32)
    if arr.ndim != 2:
        raise ValueError("Batch input must be a 2D array")
    dim = projections[0].shape[1]
    if arr.shape[1] != dim:
        raise ValueError(f"Expected vectors of dimension {dim}, received {arr.shape[1]}")
    nb = len(projections)
    bb = band_bytes(projections[0].shape[0])
    out = np.empty((arr.shape[0], nb, bb), dtype=np.uint8)
    with np.errstate(invalid="ignore", over="ignore"):     # (NaN / Inf rows: the reference computes on, NumPy only warns)
        for i in range(arr.shape[0]):
            v = arr[i]
            for b in range(nb):
                y = projections[b] @ v
                out[i, b] = np.packbits((y > 0).astype(np.uint8), bitorder="little")
    return out


def hash_batch_blas_packed(projections: Sequence[np.ndarray], vectors) -> np.ndarray:
    """Best-effort batched CPU form (one sgemm).  NOT bit-exact with the reference at sign
    boundaries (different summation order); used only as the honest "all cores" CPU line."""
    arr = np.ascontiguousarray(vectors, dtype=np.float32)
    p_all = np.concatenate([np.asarray(p, dtype=np.float32) for p in projections], axis=0)
    nb = len(projections)
    r = projections[0].shape[0]
    y = arr @ p_all.T
    bits = (y > 0).reshape(arr.shape[0], nb, r)
    return np.packbits(bits, axis=2, bitorder="little")


def exact_projection_f64(projections: Sequence[np.ndarray], vectors) -> np.ndarray:
    """float64 projections (n, num_perm): products of two f32 are exact in f64, so this is the
    true dot product to ~1e-13; used to compute sign margins, never as a parity target."""
    p_all = np.concatenate([np.asarray(p, dtype=np.float64) for p in projections], axis=0)
    return np.asarray(vectors, dtype=np.float64) @ p_all.T


class OracleHasher:
    """Interface twin of the reference ``LSHHasher`` (lshrs/hash/lsh.py:51-247) for CPU tests."""

    def __init__(self, num_bands: int, rows_per_band: int, dim: int, seed: int = 42) -> None:
        self.projections = make_projections(num_bands, rows_per_band, dim, seed)
        self.num_bands = num_bands
        self.rows_per_band = rows_per_band
        self.dim = dim

    def hash_vector(self, vector):
        from lshrs_amd._config import HashSignatures  # value type only; no compute

        return HashSignatures(hash_vector_literal(self.projections, vector, self.dim))

    def hash_batch(self, vectors):
        from lshrs_amd._config import HashSignatures

        packed = self.hash_batch_packed(vectors)
        return [HashSignatures(tuple(bytes(row[b]) for b in range(self.num_bands))) for row in packed]

    def hash_batch_packed(self, vectors) -> np.ndarray:
        arr = np.asarray(vectors, dtype=np.float32)
        if arr.ndim != 2:
            raise ValueError("Batch input must be a 2D array")
        if arr.shape[1] != self.dim:
            raise ValueError(f"Expected vectors of dimension {self.dim}, received {arr.shape[1]}")
        return hash_batch_literal_packed(self.projections, arr)


# --------------------------------------------------------------------------
# orchestrator-side vector validation
# --------------------------------------------------------------------------

def prepare_vector(vector, dim: int) -> np.ndarray:
    """lshrs/core/main.py:1075-1086 — f32, flatten, dimension check, near-zero rejection."""
    arr = np.asarray(vector, dtype=np.float32).reshape(-1)
    if arr.shape[0] != dim:
        raise ValueError(f"Vector must have dimension {dim}; received {arr.shape[0]}")
    if np.allclose(arr, 0.0, atol=1e-8):
        raise ValueError("Cannot index zero vector - norm undefined. Check embeddings for corruption.")
    return arr


def is_zero_vector_rows(vectors) -> np.ndarray:
    """Row-wise restatement of the ``np.allclose(arr, 0.0, atol=1e-8)`` test at
    lshrs/core/main.py:1083 (one bool per row; literal call per row)."""
    arr = np.asarray(vectors, dtype=np.float32)
    return np.array([bool(np.allclose(row, 0.0, atol=1e-8)) for row in arr], dtype=bool)


# --------------------------------------------------------------------------
# cosine rerank
# --------------------------------------------------------------------------

def l2_norm(vector) -> np.ndarray:
    """lshrs/utils/norm.py:48-61 — f32 flatten, np.linalg.norm, raise on 0, divide."""
    v = np.asarray(vector, dtype=np.float32).reshape(-1)
    n = np.linalg.norm(v)
    if n == 0:
        raise ValueError("Cannot normalize zero vector")
    return v / n


def cosine_similarity(query, candidates) -> np.ndarray:
    """lshrs/utils/similarity.py:80-90 — normalise query, normalise+stack candidates, sgemv."""
    q = l2_norm(query)
    rows = [l2_norm(c) for c in candidates]
    mat = np.stack(rows)  # raises ValueError on an empty candidate list, as the reference does
    return mat @ q


def top_k_cosine(query, candidates, *, k: int) -> List[Tuple[int, float]]:
    """lshrs/utils/similarity.py:157-183 — k check, scores, argpartition, argsort, python tuples."""
    if k <= 0:
        raise ValueError("k must be > 0")
    s = cosine_similarity(query, candidates)
    n = len(s)
    if n == 0:
        return []
    part = np.argpartition(-s, kth=min(k, n - 1))[:k]
    order = part[np.argsort(-s[part])]
    return [(int(i), float(s[i])) for i in order]


def rerank_batch(queries, corpus, cand_idx, k: int):
    """Batched use of top_k_cosine as ``LSHRS.query`` drives it (lshrs/core/main.py:646):
    for each query row, candidates = corpus[cand_idx[row]]."""
    out = []
    for qi in range(len(queries)):
        out.append(top_k_cosine(queries[qi], corpus[np.asarray(cand_idx[qi])], k=k))
    return out


# --------------------------------------------------------------------------
# band/row auto-configuration values observed from the reference
# --------------------------------------------------------------------------

def collision_probability(s: float, b: int, r: int) -> float:
    """P(candidate) = 1 - (1 - s^r)^b — the S-curve lshrs/utils/br.py integrates."""
    return 1.0 - (1.0 - s ** r) ** b


def false_rates(threshold: float, b: int, r: int) -> Tuple[float, float]:
    """FP = ∫_0^t P(s) ds, FN = ∫_t^1 (1 - P(s)) ds (lshrs/utils/br.py:162-220), by scipy.quad."""
    from scipy.integrate import quad

    fp, _ = quad(lambda s: collision_probability(s, b, r), 0.0, threshold)
    fn, _ = quad(lambda s: 1.0 - collision_probability(s, b, r), threshold, 1.0)
    return fp, fn


def ceil_div(a: int, b: int) -> int:
    return -(-a // b)


def expected_limit(n: int, top_p: float, top_k=None) -> int:
    """lshrs/core/main.py:650-656 — top-p cut-off used by ``query``."""
    lim = max(1, math.ceil(n * top_p))
    if top_k is not None:
        lim = min(lim, top_k)
    return lim


# --------------------------------------------------------------------------
# orchestrator ingest loop (timed as the end-to-end CPU baseline; also checks LSHRS.index in tests)
# --------------------------------------------------------------------------

def index_literal(storage, indices, vectors, projections: Sequence[np.ndarray], dim: int, buffer_size: int = 10_000) -> int:
    """lshrs/core/main.py:495-518 (index) -> :386-411 (ingest) -> :1050-1086 (_prepare_vector), lsh.py:96-134
    (hash_vector), :1113-1143 (_enqueue_operations, _flush_buffer_if_needed), :413-434 (flush): one vector at a time,
    ``num_bands`` operations appended per vector, the whole buffer handed to ``storage.batch_add`` at the first vector
    boundary where it holds ``buffer_size`` operations, and once more at the end.  Returns the number of flushes."""
    arr = np.asarray(vectors, dtype=np.float32)
    if arr.ndim != 2 or arr.shape[1] != dim:
        raise ValueError(f"Vectors must have shape (n, {dim}); received {arr.shape}")
    buffer: list = []
    flushes = 0
    for idx, vec in zip(indices, arr):
        idx = int(idx)
        if idx < 0:
            raise ValueError("index must be non-negative")
        v = prepare_vector(vec, dim)
        signatures = hash_vector_literal(projections, v, dim)
        for band_id, hash_val in enumerate(signatures):
            buffer.append((band_id, hash_val, idx))
        if len(buffer) >= buffer_size:
            storage.batch_add(list(buffer))
            buffer.clear()
            flushes += 1
    if buffer:
        storage.batch_add(list(buffer))
        flushes += 1
    return flushes


def query_literal(storage, projections: Sequence[np.ndarray], dim: int, vector, *, top_k=10, top_p=None, fetch=None):
    """lshrs/core/main.py:524-658 (query) with :1088-1111 (_candidate_counts) inlined: hash the query, read one bucket
    per band, count collisions per stored id, order by (-count, id); without ``top_p`` the first ``top_k`` ids, with it
    the candidates reranked by ``top_k_cosine`` and cut to ``max(1, ceil(n * top_p))`` (and ``top_k``)."""
    import math

    q = prepare_vector(vector, dim)
    counts: dict = {}
    for band_id, hash_val in enumerate(hash_vector_literal(projections, q, dim)):
        for candidate in storage.get_bucket(band_id, hash_val):
            counts[candidate] = counts.get(candidate, 0) + 1
    if not counts:
        return []
    ordered = sorted(counts.items(), key=lambda item: (-item[1], item[0]))
    if top_p is None:
        if top_k is None:
            top_k = len(ordered)
        if top_k <= 0:
            raise ValueError("top_k must be greater than zero when provided")
        return [idx for idx, _ in ordered[:top_k]]
    if not 0 < top_p <= 1:
        raise ValueError("top_p must be within the range (0, 1]")
    candidate_indices = [idx for idx, _ in ordered]
    arr = np.asarray(fetch(candidate_indices), dtype=np.float32)
    ranked = top_k_cosine(q, arr, k=len(candidate_indices))
    scored = [(candidate_indices[pos], score) for pos, score in ranked]
    limit = max(1, math.ceil(len(scored) * top_p))
    if top_k is not None:
        if top_k <= 0:
            raise ValueError("top_k must be greater than zero when provided")
        limit = min(limit, top_k)
    return scored[:limit]
