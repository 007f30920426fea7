"""Where a single `get_top_k` / `get_above_p` call spends its time (the one-query chain, `_query_device.OneQuery`)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lshrs_amd import LSHRS, InMemoryStorage
n, dim = 1_000_000, 768
dev = torch.device("cuda:0")
corpus = torch.randn(n, dim, device=dev, generator=torch.Generator(dev).manual_seed(20240101))
host = corpus.cpu().numpy()
idx = LSHRS(dim=dim, num_perm=256, storage=InMemoryStorage(), packed_ingest=True)
idx.index(np.arange(n, dtype=np.int64), host)
idx.set_corpus(corpus)
rng = np.random.default_rng(7)
q = host[rng.choice(n, 2000, replace=False)] + 0.1 * rng.standard_normal((2000, dim)).astype(np.float32)
for label, fn in (("get_top_k(10)", lambda v: idx.get_top_k(v, topk=10)), ("get_above_p(0.5)", lambda v: idx.get_above_p(v, p=0.5))):
    for v in q[:200]:
        fn(v)
    t0 = time.perf_counter()
    for v in q:
        fn(v)
    dt = (time.perf_counter() - t0) / len(q)
    print(f"{label}: {dt * 1e6:.1f} us per call = {1 / dt:.0f} calls/s", flush=True)
pr = cProfile.Profile(); pr.enable()
for v in q:
    idx.get_top_k(v, topk=10)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
