"""GPU parity of the cosine rerank (K2 cosine_kernel, K3 topk_kernel, l2 normalise) through the C ABI.

Tolerance (BASELINE.json north_star): |score - reference| <= 1e-5.  Order: equal to the oracle's
wherever adjacent scores differ by more than 2e-5; inside ties only the index *set* is compared
(the reference's argpartition/argsort tie order is unspecified)."""

from __future__ import annotations

import json
import os

import numpy as np
import pytest

from oracle import lshrs_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def assert_same_ranking(got, want, gap=2e-5):
    assert len(got) == len(want)
    gs = np.array([s for _, s in got])
    ws = np.array([s for _, s in want])
    assert np.abs(gs - ws).max() <= TOL if len(got) else True
    assert (np.diff(gs) <= 0).all(), "scores not descending"
    # positions must agree wherever the oracle's neighbours are clearly separated
    i = 0
    while i < len(want):
        j = i
        while j + 1 < len(want) and ws[j] - ws[j + 1] <= gap:
            j += 1
        assert {p for p, _ in got[i:j + 1]} == {p for p, _ in want[i:j + 1]}, (i, j)
        i = j + 1


def test_known_answers_of_the_reference_suite():
    from lshrs_amd import cosine_similarity, top_k_cosine

    q = np.array([1.0, 0.0, 0.0], dtype=np.float32)
    c = [np.array(v, dtype=np.float32) for v in ([1, 0, 0], [0, 1, 0], [-1, 0, 0], [1, 1, 0])]
    s = cosine_similarity(q, c)
    assert s.shape == (4,) and s.dtype == np.float32
    assert np.allclose(s, [1.0, 0.0, -1.0, 0.70710677], atol=1e-6)
    c5 = [np.array(v, dtype=np.float32) for v in ([1, .1, 0], [0, 1, 0], [1, 0, 0], [-1, 0, 0], [.9, .2, 0])]
    top = top_k_cosine(q, c5, k=3)
    assert [i for i, _ in top] == [2, 0, 4]
    assert top[0][1] == pytest.approx(1.0) and top[1][1] >= top[2][1]
    assert all(isinstance(i, int) and isinstance(v, float) for i, v in top)
    assert len(top_k_cosine(q, c5, k=10)) == 5
    with pytest.raises(ValueError, match="k must be > 0"):
        top_k_cosine(q, c5, k=0)


def test_zero_vectors_and_empty_input_raise_like_the_reference():
    from lshrs_amd import cosine_similarity, l2_norm, top_k_cosine

    q = np.array([1.0, 2.0, 3.0, 4.0], dtype=np.float32)
    good = np.ones((3, 4), dtype=np.float32)
    bad = good.copy()
    bad[1] = 0
    with pytest.raises(ValueError, match="Cannot normalize zero vector"):
        cosine_similarity(q, bad)
    with pytest.raises(ValueError, match="Cannot normalize zero vector"):
        top_k_cosine(np.zeros(4, dtype=np.float32), good, k=2)
    with pytest.raises(ValueError):
        cosine_similarity(q, [])
    with pytest.raises(ValueError, match="Cannot normalize zero vector"):
        l2_norm(np.zeros(4, dtype=np.float32))
    u = l2_norm(np.array([3.0, 4.0, 0.0], dtype=np.float32))
    assert u.dtype == np.float32 and u.shape == (3,) and np.allclose(u, [0.6, 0.8, 0.0], atol=1e-7)
    v = np.random.default_rng(0).standard_normal(1537).astype(np.float32)
    assert np.allclose(l2_norm(v), O.l2_norm(v), atol=1e-7)
    assert l2_norm(np.arange(6).reshape(2, 3)).shape == (6,)


def test_reference_goldens(golden_dir):
    from lshrs_amd import cosine_similarity, rerank_batch, top_k_cosine

    g = np.load(os.path.join(golden_dir, "g4_cosine.npz"))
    j = json.load(open(os.path.join(golden_dir, "g4_cosine.json")))
    rng = np.random.default_rng(201)
    q = rng.standard_normal(768).astype(np.float32)
    cands = rng.standard_normal((64, 768)).astype(np.float32)
    cands[5] = q * 3.0
    cands[9] = -q
    cands[11] = cands[12]
    s = cosine_similarity(q, cands)
    assert np.abs(s - g["scores_q201_c64"]).max() <= TOL
    assert s[5] == pytest.approx(1.0, abs=1e-6) and s[9] == pytest.approx(-1.0, abs=1e-6) and s[11] == s[12]
    for k, want in j["topk"].items():
        assert_same_ranking(top_k_cosine(q, cands, k=int(k)), [(i, v) for i, v in want])
    rng = np.random.default_rng(202)
    corpus = rng.standard_normal((500, 32)).astype(np.float32)
    queries = rng.standard_normal((6, 32)).astype(np.float32)
    cidx = rng.integers(0, 500, size=(6, 40))
    got = rerank_batch(queries, corpus, cidx, k=40)
    for i in range(6):
        assert_same_ranking(got[i], [(p, v) for p, v in j["batch_topk_202"][i]])


@pytest.mark.parametrize("dim,c", [(4, 7), (30, 65), (100, 1000), (768, 1000), (1536, 333), (2050, 64)])
def test_scores_and_order_vs_oracle(dim, c):
    from lshrs_amd import cosine_similarity, top_k_cosine

    rng = np.random.default_rng(dim * 7 + c)
    q = rng.standard_normal(dim).astype(np.float32)
    cands = (rng.standard_normal((c, dim)) * rng.uniform(0.01, 100, size=(c, 1))).astype(np.float32)
    s = cosine_similarity(q, cands)
    assert np.abs(s - O.cosine_similarity(q, cands)).max() <= TOL
    from oracle.build import cosine_f64

    assert np.abs(s - cosine_f64(q, cands)).max() <= 2e-6
    for k in (1, min(10, c), c, c + 5):
        assert_same_ranking(top_k_cosine(q, cands, k=k), O.top_k_cosine(q, cands, k=k))


def test_batched_gather_vs_oracle_c3_shape_sample():
    """Config 3 shape in small: 768-d corpus, 1000 candidates per query, int64 gather indices."""
    import torch

    from lshrs_amd import rerank_batch

    rng = np.random.default_rng(33)
    corpus = rng.standard_normal((20_000, 768)).astype(np.float32)
    qrows = rng.choice(20_000, 24, replace=False)
    queries = corpus[qrows] + 0.1 * rng.standard_normal((24, 768)).astype(np.float32)
    cidx = rng.integers(0, 20_000, size=(24, 1000))
    cidx[:, 0] = qrows                                      # the near-duplicate must come out on top
    d_corpus = torch.from_numpy(corpus).cuda()
    got = rerank_batch(queries, d_corpus, cidx, k=1000)
    want = O.rerank_batch(queries, corpus, cidx, k=1000)
    for i in range(24):
        assert got[i][0][0] == 0
        assert_same_ranking(got[i], want[i])
    top10 = rerank_batch(queries, d_corpus, cidx, k=10)
    for i in range(24):
        assert_same_ranking(top10[i], want[i][:10])


def test_status_codes_and_limits():
    import torch

    from lshrs_amd import rerank_batch
    from lshrs_amd.similarity import cosine_scores_device, topk_desc_device

    corpus = torch.randn(100, 64, device="cuda")
    corpus[7] = 0
    queries = torch.randn(3, 64, device="cuda")
    idx = torch.tensor([[1, 2, 3], [4, 7, 5], [6, 100, -1]], device="cuda")
    scores, status, qstatus = cosine_scores_device(corpus, queries, idx)
    assert status.cpu().tolist() == [[0, 0, 0], [0, 1, 0], [0, 2, 2]]
    assert qstatus.cpu().tolist() == [0, 0, 0]
    assert torch.isnan(scores[1, 1]) and torch.isnan(scores[2, 1])
    with pytest.raises(IndexError):
        rerank_batch(queries, corpus, torch.tensor([[1], [2], [100]], device="cuda"), k=1)
    with pytest.raises(ValueError, match="Cannot normalize zero vector"):
        rerank_batch(queries, corpus, torch.tensor([[1], [7], [3]], device="cuda"), k=1)
    # NaN scores sort last, ties by position
    sc = torch.tensor([[0.5, float("nan"), 0.5, 1.0, -1.0, float("inf"), float("-inf"), 0.0, -0.0]], device="cuda")
    order, vals = topk_desc_device(sc, 9)
    assert order.cpu().tolist()[0][:4] == [5, 3, 0, 2]
    assert order.cpu().tolist()[0][-1] == 1 and order.cpu().tolist()[0][-2] == 6
    big = torch.randn(2, 16384, device="cuda")
    order, vals = topk_desc_device(big, 16384)
    ref = torch.sort(big, dim=1, descending=True, stable=True)
    assert torch.equal(vals, ref.values)
    assert torch.equal(order.long(), ref.indices)
    # lists longer than one LDS network go through the global-memory network: same order
    for cbig, kk in ((16385, 16385), (40_000, 100), (300_001, 300_001)):
        big = torch.randn(3, cbig, device="cuda")
        big[1, 7] = float("nan")
        big[2, :5] = 10.0                                    # ties (and the maximum) -> ascending position
        order, vals = topk_desc_device(big, kk)
        ref = torch.sort(torch.nan_to_num(big, nan=float("-inf")), dim=1, descending=True, stable=True)
        assert torch.equal(order.long()[0], ref.indices[0, :kk]) and torch.equal(vals[0], ref.values[0, :kk])
        assert order[2, :5].tolist() == [0, 1, 2, 3, 4]
        if kk == cbig:
            assert int(order[1, -1]) == 7 and bool(torch.isnan(vals[1, -1]))
            assert torch.equal(torch.sort(order.long(), dim=1).values, torch.arange(cbig, device="cuda").expand(3, -1))


def test_full_size_config3_properties():
    """BASELINE config 3: 1M x 768 corpus on the device, 10k queries x 1k candidates."""
    import torch

    from lshrs_amd import rerank_batch

    gen = torch.Generator("cuda").manual_seed(7)
    corpus = torch.randn(1_000_000, 768, device="cuda", generator=gen)
    qrows = torch.randperm(1_000_000, device="cuda", generator=gen)[:10_000]
    queries = corpus[qrows] + 0.1 * torch.randn(10_000, 768, device="cuda", generator=gen)
    cidx = torch.randint(0, 1_000_000, (10_000, 1000), device="cuda", generator=gen)
    cidx[:, 17] = qrows
    order, scores = rerank_batch(queries, corpus, cidx, k=1000, return_tensors=True)
    assert order.shape == (10_000, 1000) and scores.shape == (10_000, 1000)
    assert bool((scores[:, :-1] >= scores[:, 1:]).all()), "not sorted"
    assert bool((order[:, 0] == 17).all()), "planted near-duplicate not ranked first"
    assert bool((scores[:, 0] > 0.99).all())
    # (a random candidate list holds the query's own row a second time for ~0.1 % of the queries)
    assert (scores[:, 1] < 0.3).float().mean().item() > 0.99
    assert bool((torch.sort(order.long(), dim=1).values == torch.arange(1000, device="cuda")).all()), "not a permutation"
    assert bool((scores.abs() <= 1.0 + 1e-6).all())
    # CPU check on sampled queries
    pick = [0, 1234, 9999]
    q_h, c_h = queries[pick].cpu().numpy(), cidx[pick].cpu().numpy()
    for t, qi in enumerate(pick):
        cand = corpus[cidx[qi]].cpu().numpy()
        want = O.top_k_cosine(q_h[t], cand, k=1000)
        got = list(zip(order[qi].cpu().tolist(), scores[qi].cpu().tolist()))
        assert_same_ranking(got, want)
