"""Timeline of the pipelined LSHRS.index(): per chunk, when its grouping was enqueued, how long the enqueue took, when the
finisher started / ended; and the unit's hash call.   python tools/e2e_profile2.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lshrs_amd import LSHRS, InMemoryStorage, packed_ops, _ingest

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
if len(sys.argv) > 2:
    _ingest.CsrIngest.chunk_rows = int(sys.argv[2])
host = np.random.default_rng(1).standard_normal((rows, 768), dtype=np.float32)
ids = np.arange(rows, dtype=np.int64)
idx = LSHRS(dim=768, num_perm=256, storage=InMemoryStorage(), packed_ingest=True)
idx.index(ids[:100_000], host[:100_000])
log = []
T0 = [0.0]
now = lambda: (time.perf_counter() - T0[0]) * 1e3
orig_init, orig_finish = packed_ops.DeviceCSRJob.__init__, packed_ops.DeviceCSRJob.finish
def init(self, ids_, keys_dev, ids_dev=None):
    t = now(); orig_init(self, ids_, keys_dev, ids_dev); log.append(("enqueue", t, now(), len(ids_)))
def finish(self):
    t = now(); self.event.synchronize(); t1 = now(); out = orig_finish(self); log.append(("finish", t, t1, now()))
    return out
packed_ops.DeviceCSRJob.__init__ = init
packed_ops.DeviceCSRJob.finish = finish
h = idx._hasher
orig_hash = h.hash_batch_packed
def hb(*a, **k):
    t = now(); out = orig_hash(*a, **k); log.append(("hash_batch_packed", t, now())); return out
h.hash_batch_packed = hb
for rep in range(3):
    log.clear()
    T0[0] = time.perf_counter()
    idx.index(ids + 10_000_000 * (rep + 1), host)
    total = now()
    print(f"--- index(): {total:.1f} ms = {rows / total / 1e3:.2f} M vec/s")
    for e in sorted(log, key=lambda e: e[1]):
        print("   ", e[0], " ".join(f"{v:.1f}" if isinstance(v, float) else str(v) for v in e[1:]))
