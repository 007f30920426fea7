"""Cosine rerank on MI355X — ``l2_norm``, ``cosine_similarity``, ``top_k_cosine`` and the
batched ``rerank_batch``.

Function names, argument meaning, return types and error behaviour follow
lshrs/utils/similarity.py:26-183 and lshrs/utils/norm.py:4-61; the arithmetic runs in
``cosine_kernel`` / ``topk_kernel`` of ``csrc/rerank.hip`` (C ABI:
``lshrs_cosine_batch_f32`` / ``lshrs_topk_desc_f32``).  No CPU compute path.

Numerics: the kernel evaluates ``dot(c, q) / (||c|| * ||q||)`` in float32 with a fixed
per-lane + wave-tree summation order; the reference normalises first and then takes the
dot product.  Both are within ~1e-7 of the real cosine; the contract is |Δ| <= 1e-5.
Ordering: descending score, ties by ascending candidate position, NaN last (the
reference's argpartition/argsort order on ties is unspecified).
"""

from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native

__all__ = ["l2_norm", "cosine_similarity", "top_k_cosine", "rerank_batch", "rerank_padded", "cosine_scores_device",
           "l2_normalize_device",
           "topk_desc_device"]

_TOPK_MAX_Q = 65535


def _as_matrix(candidates, dim: Optional[int] = None) -> np.ndarray:
    """Sequence of vectors / 2-D array -> contiguous (C, dim) float32 (each row flattened,
    as ``l2_norm`` would flatten it, norm.py:48)."""
    if isinstance(candidates, np.ndarray) and candidates.ndim == 2:
        return np.ascontiguousarray(candidates, dtype=np.float32)
    rows = [np.asarray(c, dtype=np.float32).reshape(-1) for c in candidates]
    if not rows:
        # the reference reaches np.stack([]) here (similarity.py:85)
        raise ValueError("need at least one array to stack")
    return np.ascontiguousarray(np.stack(rows))


def cosine_scores_device(corpus, queries, cand_idx=None, *, c: Optional[int] = None):
    """Device-level entry: tensors in, tensors out.

    corpus (m, dim) f32, queries (q, dim) f32, cand_idx (q, c) int64 or None (then the candidates
    of query i are corpus rows [i*c, (i+1)*c)).  Returns (scores (q, c) f32, status (q, c) u8,
    qstatus (q,) u8) on the same device; see include/lshrs_hip.h for the status codes.
    """
    torch = _native.require_gpu()
    lib = _native.load()
    if corpus.dtype != torch.float32 or queries.dtype != torch.float32:
        raise TypeError("corpus and queries must be float32 tensors")
    if corpus.dim() != 2 or queries.dim() != 2 or corpus.shape[1] != queries.shape[1]:
        raise ValueError("corpus must be (m, dim) and queries (q, dim)")
    if corpus.stride(1) != 1:
        corpus = corpus.contiguous()
    queries = queries.contiguous()
    dev = corpus.device
    q = int(queries.shape[0])
    if cand_idx is not None:
        if cand_idx.dtype != torch.int64 or cand_idx.dim() != 2 or cand_idx.shape[0] != q:
            raise ValueError("cand_idx must be an int64 tensor of shape (q, c)")
        cand_idx = cand_idx.contiguous()
        c = int(cand_idx.shape[1])
    elif c is None:
        raise ValueError("give cand_idx or c")
    scores = torch.empty((q, c), dtype=torch.float32, device=dev)
    status = torch.empty((q, c), dtype=torch.uint8, device=dev)
    qstatus = torch.empty((q,), dtype=torch.uint8, device=dev)
    if q == 0 or c == 0:
        return scores, status, qstatus
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _native.check(
            lib.lshrs_cosine_batch_f32(corpus.data_ptr(), corpus.shape[0], corpus.stride(0), corpus.shape[1],
                                       queries.data_ptr(), q, cand_idx.data_ptr() if cand_idx is not None else None,
                                       c, scores.data_ptr(), status.data_ptr(), qstatus.data_ptr(), stream),
            "lshrs_cosine_batch_f32")
    return scores, status, qstatus


def topk_desc_device(scores, k: int):
    """(q, c) f32 scores -> (order (q, k) int32 positions, sorted (q, k) f32), descending."""
    torch = _native.require_gpu()
    lib = _native.load()
    scores = scores.contiguous()
    q, c = int(scores.shape[0]), int(scores.shape[1])
    if k > c or k < 0:
        raise ValueError("k must be within [0, c]")
    order = torch.empty((q, k), dtype=torch.int32, device=scores.device)
    sorted_scores = torch.empty((q, k), dtype=torch.float32, device=scores.device)
    if q == 0 or k == 0:
        return order, sorted_scores
    with torch.cuda.device(scores.device):
        stream = torch.cuda.current_stream(scores.device).cuda_stream
        for lo in range(0, q, _TOPK_MAX_Q):      # (the global-memory network takes <= 65535 queries per call)
            hi = min(q, lo + _TOPK_MAX_Q)
            nbytes = int(lib.lshrs_topk_workspace_bytes(hi - lo, c))
            if nbytes < 0:
                _native.check(nbytes, "lshrs_topk_workspace_bytes")
            ws = torch.empty(nbytes, dtype=torch.uint8, device=scores.device) if nbytes else None
            _native.check(lib.lshrs_topk_desc_f32(scores[lo:hi].data_ptr(), hi - lo, c, k, order[lo:hi].data_ptr(),
                                                  sorted_scores[lo:hi].data_ptr(), ws.data_ptr() if ws is not None else None,
                                                  stream), "lshrs_topk_desc_f32")
    return order, sorted_scores


def _raise_for_status(status, qstatus) -> None:
    if bool((qstatus != 0).any()) or bool((status == 1).any()):
        raise ValueError("Cannot normalize zero vector")
    if bool((status == 2).any()):
        raise IndexError("candidate index out of range of the corpus")


def l2_normalize_device(x):
    """Row-wise ``x / ||x||`` for a device (n, dim) float32 tensor -> (out, status)."""
    torch = _native.require_gpu()
    lib = _native.load()
    if x.dtype != torch.float32 or x.dim() != 2 or not x.is_cuda:
        raise TypeError("l2_normalize_device expects a float32 device tensor of shape (n, dim)")
    if x.stride(1) != 1:
        x = x.contiguous()
    out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
    status = torch.empty((x.shape[0],), dtype=torch.uint8, device=x.device)
    if x.shape[0]:
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            _native.check(lib.lshrs_l2_normalize_f32(x.data_ptr(), x.shape[0], x.stride(0), x.shape[1],
                                                     out.data_ptr(), status.data_ptr(), stream),
                          "lshrs_l2_normalize_f32")
    return out, status


def _upload(torch, a: np.ndarray):
    """Host array -> device tensor.  The array is only read: a read-only view (``np.frombuffer``, a memory map, a
    broadcast) is uploaded as it is, without the copy ``torch.from_numpy`` asks for in its warning."""
    import warnings

    a = np.ascontiguousarray(a)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        return torch.from_numpy(a).cuda()


def l2_norm(vector) -> np.ndarray:
    """Unit-length copy of ``vector`` (float32, flattened); zero vector -> ValueError (norm.py:48-61)."""
    torch = _native.require_gpu()
    vec = np.ascontiguousarray(np.asarray(vector, dtype=np.float32).reshape(-1))
    if vec.size == 0:
        raise ValueError("Cannot normalize zero vector")
    out, status = l2_normalize_device(_upload(torch, vec).reshape(1, -1))
    if int(status[0]) != 0:
        raise ValueError("Cannot normalize zero vector")
    return out.reshape(-1).cpu().numpy()


def cosine_similarity(query, candidates) -> np.ndarray:
    """Cosine of ``query`` against every candidate -> float32 array (similarity.py:26-90)."""
    torch = _native.require_gpu()
    q = np.asarray(query, dtype=np.float32).reshape(-1)
    mat = _as_matrix(candidates)
    if mat.shape[1] != q.shape[0]:
        raise ValueError(f"shapes {mat.shape} and {q.shape} not aligned")
    dev_c = _upload(torch, mat)
    dev_q = _upload(torch, q).reshape(1, -1)
    scores, status, qstatus = cosine_scores_device(dev_c, dev_q, None, c=mat.shape[0])
    _raise_for_status(status, qstatus)
    return scores.reshape(-1).cpu().numpy()


def top_k_cosine(query, candidates, *, k: int) -> List[Tuple[int, float]]:
    """The ``k`` most similar candidates as ``[(position, score), ...]`` in descending score
    (similarity.py:93-183).  ``k <= 0`` -> ValueError; ``k > len(candidates)`` returns them all."""
    if k <= 0:
        raise ValueError("k must be > 0")
    torch = _native.require_gpu()
    q = np.asarray(query, dtype=np.float32).reshape(-1)
    mat = _as_matrix(candidates)
    if mat.shape[1] != q.shape[0]:
        raise ValueError(f"shapes {mat.shape} and {q.shape} not aligned")
    n = mat.shape[0]
    dev_c = _upload(torch, mat)
    dev_q = _upload(torch, q).reshape(1, -1)
    scores, status, qstatus = cosine_scores_device(dev_c, dev_q, None, c=n)
    _raise_for_status(status, qstatus)
    order, sorted_scores = topk_desc_device(scores, min(k, n))
    pos = order.reshape(-1).cpu().numpy()
    val = sorted_scores.reshape(-1).cpu().numpy()
    return [(int(p), float(v)) for p, v in zip(pos, val)]


def rerank_batch(queries, corpus, cand_idx, *, k: int, return_tensors: bool = False):
    """Batched rerank: for query i, score ``corpus[cand_idx[i]]`` and order descending.

    ``queries`` (q, dim), ``corpus`` (m, dim), ``cand_idx`` (q, c) may be NumPy arrays or device
    tensors (a device-resident corpus is reused across calls).  Returns, per query, the list
    ``[(position within cand_idx[i], score)]`` of length min(k, c) — the same thing a loop of
    ``top_k_cosine(queries[i], corpus[cand_idx[i]], k=k)`` returns — or the two device tensors
    ``(order, scores)`` when ``return_tensors`` is set.
    """
    if k <= 0:
        raise ValueError("k must be > 0")
    torch = _native.require_gpu()

    def dev(a, dtype):
        if isinstance(a, torch.Tensor):
            return a.cuda() if not a.is_cuda else a
        return _upload(torch, np.asarray(a, dtype=dtype))

    d_q = dev(queries, np.float32)
    d_c = dev(corpus, np.float32)
    d_i = dev(cand_idx, np.int64)
    scores, status, qstatus = cosine_scores_device(d_c, d_q, d_i)
    _raise_for_status(status, qstatus)
    order, sorted_scores = topk_desc_device(scores, min(k, int(d_i.shape[1])))
    if return_tensors:
        return order, sorted_scores
    o = order.cpu().numpy()
    s = sorted_scores.cpu().numpy()
    return [[(int(p), float(v)) for p, v in zip(o[i], s[i])] for i in range(o.shape[0])]


def rerank_padded_arrays(queries, corpus, cand_idx):
    """Ragged batched rerank, arrays out: ``cand_idx`` is ``(q, c_max)`` int64 with ``-1`` padding after each query's
    candidates.  Returns ``(order, scores)``, both ``(q, c_max)`` NumPy arrays: ``order[i, j]`` = position (column of
    ``cand_idx``) of query i's j-th best candidate, ``scores[i, j]`` its cosine, descending; padding scores NaN on the
    device and sorts last, so row i is meaningful up to its number of valid candidates.  Zero-norm vectors raise like
    the reference; an index outside the corpus raises ``IndexError``."""
    torch = _native.require_gpu()

    def dev(a, dtype):
        if isinstance(a, torch.Tensor):
            return a if a.is_cuda else a.cuda()
        return _upload(torch, np.asarray(a, dtype=dtype))

    d_q, d_c, d_i = dev(queries, np.float32), dev(corpus, np.float32), dev(cand_idx, np.int64)
    scores, status, qstatus = cosine_scores_device(d_c, d_q, d_i)
    valid = d_i >= 0
    if bool((qstatus != 0).any()) or bool(((status == 1) & valid).any()):
        raise ValueError("Cannot normalize zero vector")
    if bool(((status == 2) & valid).any()):
        raise IndexError("candidate index out of range of the corpus")
    order, sorted_scores = topk_desc_device(scores, int(d_i.shape[1]))
    return order.cpu().numpy(), sorted_scores.cpu().numpy()


def rerank_padded(queries, corpus, cand_idx):
    """:func:`rerank_padded_arrays` as Python objects: per query ``[(position, score), ...]`` over its valid candidates,
    descending."""
    order, scores = rerank_padded_arrays(queries, corpus, cand_idx)
    idx = cand_idx.cpu().numpy() if hasattr(cand_idx, "cpu") else np.asarray(cand_idx)
    counts = (idx >= 0).sum(axis=1)
    return [[(int(p), float(v)) for p, v in zip(order[i, :counts[i]], scores[i, :counts[i]])]
            for i in range(order.shape[0])]
