"""What moves `cosine_kernel` between 4.85 and 5.31 ms per 10k x 1k pass from box to box / run to run?  

One box, one process: the pass as bench.py measures it (4 warm-ups, median of 9) under different histories -
  cold        straight after the corpus fill (what `bench.py --only rerank` and the rocprof passes of round 4 measured)
  after-idle  after 3 s of nothing
  after-load  directly behind 2 s of the signature pass (what the full bench line measures: the rerank comes last)
  long        200 passes back to back: first / median / last ten
  sorted      the same candidates, each query's list sorted by row (locality of the gather)
  fresh-alloc the corpus copied into a freshly allocated tensor
with `rocm-smi` clocks and power sampled from a second thread during the long run.

    python tools/rerank_repro.py
"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lshrs_amd import LSHHasher
from lshrs_amd.similarity import cosine_scores_device, topk_desc_device

dev = torch.device("cuda:0")
m, dim, q, c = 1_000_000, 768, 10_000, 1_000


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    keep = [l.strip() for l in out.splitlines() if re.search(r"Power|sclk|mclk|fclk|socclk", l)]
    return " | ".join(re.sub(r"\s+", " ", k) for k in keep)


rng = np.random.default_rng(20240101)
corpus = torch.empty((m, dim), dtype=torch.float32, device=dev)
for lo in range(0, m, 50_000):
    corpus[lo:lo + 50_000] = torch.from_numpy(rng.standard_normal((50_000, dim), dtype=np.float32)).to(dev)
rng7, rng8 = np.random.default_rng(7), np.random.default_rng(8)
qrows = torch.from_numpy(rng7.choice(m, q, replace=False)).to(dev)
queries = corpus[qrows] + torch.from_numpy((0.1 * rng7.standard_normal((q, dim))).astype(np.float32)).to(dev)
cidx = torch.from_numpy(rng8.integers(0, m, (q, c), dtype=np.int64)).to(dev)
bytes_per_launch = (4.0 * dim + 8 + 4) * q * c


def passes(n, cor=None, idx=None, warm=0):
    cor = corpus if cor is None else cor
    idx = cidx if idx is None else idx
    for _ in range(warm):
        cosine_scores_device(cor, queries, idx)
    torch.cuda.synchronize()
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        s = cosine_scores_device(cor, queries, idx)
        b.record()
        topk_desc_device(s[0], c)
        evs.append((a, b))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in evs]


def report(name, ms):
    med = sorted(ms)[len(ms) // 2]
    print(f"{name:12s} median {med:.3f} ms = {bytes_per_launch / med / 1e6 / 8000:.3f} of 8 TB/s   first {ms[0]:.3f}  min {min(ms):.3f}  max {max(ms):.3f}  n {len(ms)}", flush=True)


print("idle:", smi(), flush=True)
report("cold", passes(9, warm=4))
time.sleep(3.0)
report("after-idle", passes(9, warm=4))
h = LSHHasher(16, 16, dim, seed=42, audit_every=0)
keys = h.hash_device(corpus)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.0:
    for _ in range(20):
        h.hash_device(corpus, out=keys)
report("after-load", passes(9, warm=4))
time.sleep(3.0)
samples = []
stop = False


def sampler():
    while not stop:
        samples.append(smi())
        time.sleep(0.25)


t = threading.Thread(target=sampler)
t.start()
long = passes(200)
stop = True
t.join()
report("long", long)
print("   long: first ten", " ".join(f"{v:.2f}" for v in long[:10]), "| last ten", " ".join(f"{v:.2f}" for v in long[-10:]))
for s in samples[:2] + samples[-2:]:
    print("   during long:", s)
report("sorted", passes(9, idx=torch.sort(cidx, dim=1)[0], warm=4))
junk = [torch.empty(int(s), dtype=torch.uint8, device=dev) for s in (3e8, 7e8, 1.1e9, 5e8)]
fresh = torch.empty_like(corpus)
fresh.copy_(corpus)
del junk
report("fresh-alloc", passes(9, cor=fresh, warm=4))
report("cold again", passes(9, warm=4))
print("idle:", smi(), flush=True)

# ---- the alternation: is it WHICH output block the pass writes?  The same launch through the C ABI into fixed buffers.
from lshrs_amd import _native
lib = _native.load()
stream = torch.cuda.current_stream(dev).cuda_stream


def fixed(scores, status, qstatus, n=20):
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _native.check(lib.lshrs_cosine_batch_f32(corpus.data_ptr(), m, corpus.stride(0), dim, queries.data_ptr(), q, cidx.data_ptr(), c,
                                                 scores.data_ptr(), status.data_ptr(), qstatus.data_ptr(), stream), "cosine")
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in evs]


bufs = [(torch.empty((q, c), dtype=torch.float32, device=dev), torch.empty((q, c), dtype=torch.uint8, device=dev),
         torch.empty((q,), dtype=torch.uint8, device=dev)) for _ in range(4)]
for i, b in enumerate(bufs):
    ms = fixed(*b)
    print(f"fixed output buffers #{i} (scores at {b[0].data_ptr():#x}): " + " ".join(f"{v:.2f}" for v in ms[:12]), flush=True)
ms = []
for k in range(12):
    ms += fixed(*bufs[k % 2], n=1)
print("alternating #0 / #1:", " ".join(f"{v:.2f}" for v in ms))
ms = passes(12)
print("through the Python entry (allocates its outputs):", " ".join(f"{v:.2f}" for v in ms))
# kernel alone vs kernel + top-k in between (what the bench pass does)
ev = []
for k in range(12):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    s = cosine_scores_device(corpus, queries, cidx)
    b.record()
    ev.append((a, b))
torch.cuda.synchronize()
print("Python entry, no top-k between the passes:", " ".join(f"{a.elapsed_time(b):.2f}" for a, b in ev))
