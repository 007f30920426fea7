"""LSHRS.query_many end to end through the public API: 1 M x 768 corpus indexed with packed ingest, then
batches of queries (noisy copies of corpus rows, as in config 3): top-k by collisions, and top-p with the cosine rerank on
a device-resident corpus.  Prints the time of each stage of one call beside the per-query loop (`query`)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lshrs_amd import LSHRS, InMemoryStorage

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
dim = 768
dev = torch.device("cuda:0")
corpus = torch.randn(n, dim, device=dev, generator=torch.Generator(dev).manual_seed(20240101))
host = corpus.cpu().numpy()
idx = LSHRS(dim=dim, num_perm=256, storage=InMemoryStorage(), packed_ingest=True, vector_fetch_fn=lambda ids: host[np.asarray(ids)])
t0 = time.perf_counter(); idx.index(np.arange(n, dtype=np.int64), host); t_index = time.perf_counter() - t0
print(f"index {n} rows: {t_index * 1e3:.1f} ms ({n / t_index / 1e6:.2f} M vec/s)", flush=True)
rng = np.random.default_rng(7)
rows = rng.choice(n, nq, replace=False)
q = host[rows] + 0.1 * rng.standard_normal((nq, dim)).astype(np.float32)
for label, kw in (("top_k=10 (collisions only)", dict(top_k=10)), ("top_p=0.5 (rerank, device corpus)", dict(top_k=None, top_p=0.5, corpus=corpus)),
                  ("top_p=0.5, top_k=10 (rerank, device corpus)", dict(top_k=10, top_p=0.5, corpus=corpus))):
    idx.query_many(q[:100], **kw)
    best = None
    for _ in range(3):
        t0 = time.perf_counter(); out = idx.query_many(q, **kw); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    cands = sum(len(o) for o in out)
    hit = np.mean([len(o) > 0 and (o[0] if not isinstance(o[0], tuple) else o[0][0]) == r for o, r in zip(out, rows)])
    print(f"query_many {label}: {best * 1e3:.1f} ms for {nq} queries = {nq / best:.0f} q/s; results {cands}; source row first: {hit:.3f}", flush=True)
t0 = time.perf_counter()
for v in q[:300]:
    idx.get_top_k(v, topk=10)
dt = (time.perf_counter() - t0) / 300
print(f"get_top_k per call: {dt * 1e6:.0f} us = {1 / dt:.0f} q/s", flush=True)
pr = cProfile.Profile(); pr.enable(); idx.query_many(q, top_k=None, top_p=0.5, corpus=corpus); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
