#!/usr/bin/env python3
"""End-to-end rates through the public API on this box (host-resident input, in-memory storage):
hash_batch_packed (PCIe-inclusive), LSHRS.index via operation tuples vs packed arrays, query loop vs query_many."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lshrs_amd import LSHRS, InMemoryStorage, LSHHasher

rng = np.random.default_rng(0)
n, dim = 400_000, 768
x = rng.standard_normal((n, dim)).astype(np.float32)
out = {}
h = LSHHasher(16, 16, dim)
h.hash_batch_packed(x[:50_000])
t = time.perf_counter(); keys = h.hash_batch_packed(x); dt = time.perf_counter() - t
out["hash_batch_packed_host_in_host_out_vec_per_s"] = n / dt
xp = torch.from_numpy(x).pin_memory()
t = time.perf_counter(); xd = xp.cuda(non_blocking=True); torch.cuda.synchronize(); dt_h2d = time.perf_counter() - t
out["h2d_pinned_GBps"] = x.nbytes / dt_h2d / 1e9
for label, packed, m in (("index_op_tuples", False, 100_000), ("index_packed_arrays", True, 400_000)):
    idx = LSHRS(dim=dim, num_perm=256, storage=InMemoryStorage(), packed_ingest=packed, buffer_size=160_000)
    idx.index(list(range(1000)), x[:1000])
    t = time.perf_counter(); idx.index(list(range(1000, 1000 + m)), x[:m]); dt = time.perf_counter() - t
    out[label + "_vec_per_s"] = m / dt
idx = LSHRS(dim=dim, num_perm=256, storage=InMemoryStorage(), packed_ingest=True, vector_fetch_fn=lambda ids: x[np.asarray(ids)])
idx.index(list(range(n)), x)
q = x[rng.choice(n, 2000, replace=False)] + 0.02 * rng.standard_normal((2000, dim)).astype(np.float32)
idx.query_many(q[:50], top_k=None, top_p=1.0)
t = time.perf_counter(); a = [idx.query(v, top_k=None, top_p=1.0) for v in q[:300]]; dt_loop = (time.perf_counter() - t) / 300
t = time.perf_counter(); b = idx.query_many(q, top_k=None, top_p=1.0); dt_many = (time.perf_counter() - t) / 2000
assert [[i for i, _ in r] for r in a] == [[i for i, _ in r] for r in b[:300]]
out["query_loop_ms_per_query"] = 1e3 * dt_loop
out["query_many_ms_per_query"] = 1e3 * dt_many
t = time.perf_counter(); idx.get_top_k(q[0], topk=10); out["get_top_k_single_ms"] = 1e3 * (time.perf_counter() - t)
print(json.dumps(out))
