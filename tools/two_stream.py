"""Does a second stream recover stage 1's tail?  1 M rows = 3 906.25 workgroups of 256 rows on 256 CUs = 15.26 rounds: the
last round runs on a quarter of the chip, and stage 2 (0.07 ms) waits behind it.  With consecutive batches on alternating
streams the next batch's workgroups can take the CUs the tail leaves idle.

    python tools/two_stream.py [seconds]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
dev = torch.device("cuda:0")
n, dim = 1_000_000, 768
x = torch.randn(n, dim, device=dev, generator=torch.Generator(dev).manual_seed(20240101))
h = LSHHasher(16, 16, dim, seed=42)
ref = h.hash_device(x).clone()
keys = [torch.empty_like(ref) for _ in range(3)]
streams = [torch.cuda.Stream(dev) for _ in range(3)]


def run(mode, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "sync":
        for i in range(steps):
            h.hash_device(x, out=keys[0])
    else:
        k = {"async-1": 1, "async-2": 2, "async-3": 3}[mode]
        handles = []
        for i in range(steps):
            with torch.cuda.stream(streams[i % k]):
                handles.append(h.hash_device_async(x, out=keys[i % k]))
            if len(handles) > 2:
                handles.pop(0).result()
        for hd in handles:
            hd.result()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


for mode in ("sync", "async-1", "async-2", "async-3", "sync", "async-2"):
    run(mode, 30)
    per = run(mode, 50)
    steps = max(100, int(seconds / per))
    per = run(mode, steps)
    ok = all(torch.equal(k, ref) for k in keys[:1])
    print(f"{mode:8s} {steps:5d} steps  {per * 1e3:.4f} ms/step  {n / per / 1e6:7.1f} M vec/s  keys ok: {ok}", flush=True)
