#!/usr/bin/env python3
"""Same process, same box: the synchronous step with the host polling the pinned epoch word (default) against the stream
wait (`_poll_spins = 0`), interleaved; and the host-engine route (tie_replay="off", proven windows) chunk-pipelined against
one pass + engine."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher

n, dim = 1_000_000, 768
x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(1000))
h = LSHHasher(16, 16, dim, seed=42)
keys = h.hash_device(x)
for _ in range(60):
    h.hash_device(x, out=keys)


def run(spins, steps=300):
    h._poll_spins = spins
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        h.hash_device(x, out=keys)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


for rep in range(3):
    print(f"rep {rep}: poll {run(400_000):.4f} ms/step, stream wait {run(0):.4f} ms/step", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "host":
    for chunk in (262_144, 10 ** 9):
        hh = LSHHasher(16, 16, dim, seed=42, tie_replay="off")
        hh.pipeline_chunk_rows = chunk
        k2 = hh.hash_device(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            k2 = hh.hash_device(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"host engine, proven windows, chunk rows {chunk}: {1e3 * dt:.1f} ms/step = {n / dt / 1e6:.2f} M vec/s, equal to default: "
              f"{bool(torch.equal(k2, keys))}, stats {dict((k, v) for k, v in hh.last_stats.items() if not k.startswith('t_'))}", flush=True)
