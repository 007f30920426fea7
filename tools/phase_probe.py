#!/usr/bin/env python3
"""Where the host side of a synchronous hash_device step goes: the library call that enqueues the three kernels, the wait for
the stream, the rest of the interpreter's work - per shape.  (developer tool)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lshrs_amd import LSHHasher, _native

lib = _native.load()
real_call = lib.lshrs_sig_hash_batch_split_replay_f32
real_sync = lib.lshrs_stream_synchronize
acc = {"enqueue": 0.0, "wait": 0.0}


class Wrap:
    def __init__(self, fn, key):
        self.fn, self.key = fn, key

    def __call__(self, *a):
        t = time.perf_counter()
        r = self.fn(*a)
        acc[self.key] += time.perf_counter() - t
        return r


lib.lshrs_sig_hash_batch_split_replay_f32 = Wrap(real_call, "enqueue")
lib.lshrs_stream_synchronize = Wrap(real_sync, "wait")
for nb, r, dim, n in ((20, 6, 128, 1_000_000), (16, 16, 768, 1_000_000), (20, 6, 128, 512)):
    h = LSHHasher(nb, r, dim, seed=42)
    x = torch.randn(n, dim, device="cuda")
    keys = h.hash_device(x).clone()
    for _ in range(100):
        h.hash_device(x, out=keys)
    acc["enqueue"] = acc["wait"] = 0.0
    N = 300
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        h.hash_device(x, out=keys)
    tot = (time.perf_counter() - t0) / N
    print(f"{nb} x {r} x {dim}, n = {n}: step {tot * 1e6:.1f} us = enqueue (C call, 3 launches) {acc['enqueue'] / N * 1e6:.1f} + stream wait "
          f"{acc['wait'] / N * 1e6:.1f} + interpreter {(tot - (acc['enqueue'] + acc['wait']) / N) * 1e6:.1f}", flush=True)
