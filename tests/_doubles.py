"""Test doubles: an oracle-backed hasher with the interface ``LSHRS`` consumes.

Lives under tests/ on purpose — the product package never imports the oracle.  CPU-only tests use
it to exercise the orchestrator's host logic (buffering, flush boundaries, error timing, query
ordering) where no GPU exists; the GPU tests run the same assertions through the HIP hasher.
"""

from __future__ import annotations

import numpy as np

from oracle import lshrs_oracle as O


class OracleBackedHasher(O.OracleHasher):
    def hash_batch_packed(self, vectors, *, return_row_flags: bool = False, **_):
        arr = np.asarray(vectors, dtype=np.float32)
        keys = super().hash_batch_packed(arr)
        if not return_row_flags:
            return keys
        flags = O.is_zero_vector_rows(arr).astype(np.uint8)
        flags |= (np.isnan(arr).any(axis=1).astype(np.uint8) << 1)
        return keys, flags


def make_cpu_lshrs(monkeypatch, **kw):
    """LSHRS whose hasher and reranker are the oracle (CPU): host logic only."""
    import lshrs_amd.core as core
    from lshrs_amd import LSHRS, InMemoryStorage
    from lshrs_amd.bandrows import get_optimal_config

    dim = kw["dim"]
    num_perm = kw.get("num_perm", 128)
    nb, r = kw.get("num_bands"), kw.get("rows_per_band")
    if nb is None or r is None:
        nb, r = get_optimal_config(num_perm, kw.get("similarity_threshold", 0.5))
    kw.setdefault("storage", InMemoryStorage())
    kw["hasher"] = OracleBackedHasher(nb, r, dim, kw.get("seed", 42))
    monkeypatch.setattr(core, "top_k_cosine", O.top_k_cosine)
    monkeypatch.setattr(core, "_rerank_padded", oracle_rerank_padded)
    from lshrs_amd import packed_ops

    # (the device counting sort needs a GPU; the host grouping builds the same structure)
    monkeypatch.setattr(core, "_bucket_csr", lambda ids, keys: packed_ops._csr_host(
        np.ascontiguousarray(np.asarray(ids, dtype=np.int64)), np.ascontiguousarray(keys, dtype=np.uint8)))
    return LSHRS(**kw)


def oracle_rerank_padded(queries, corpus, cand_idx):
    """CPU stand-in for lshrs_amd.similarity.rerank_padded_arrays: the oracle's top_k_cosine per query, as the
    ``(order, scores)`` matrices (rows padded with position 0 / NaN behind the valid candidates)."""
    corpus = np.asarray(corpus, dtype=np.float32)
    cand_idx = np.asarray(cand_idx)
    order = np.zeros(cand_idx.shape, dtype=np.int32)
    scores = np.full(cand_idx.shape, np.nan, dtype=np.float32)
    for qi in range(len(queries)):
        idx = cand_idx[qi]
        idx = idx[idx >= 0]
        if len(idx):
            ranked = O.top_k_cosine(queries[qi], corpus[idx], k=len(idx))
            order[qi, :len(ranked)] = [p for p, _ in ranked]
            scores[qi, :len(ranked)] = [v for _, v in ranked]
    return order, scores
