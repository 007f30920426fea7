"""The synchronous step on short shapes: ms per `hash_device` step at 1 M rows with the pinned-word poll
(`spin_wait_us`) on and off, and the host's share (step - the kernels' own durations, HIP events riding on the dispatches)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher
dev = torch.device("cuda:0")
n = 1_000_000
for nb, r, dim in ((16, 4, 128), (20, 6, 128), (16, 16, 768)):
    x = torch.randn(n, dim, device=dev, generator=torch.Generator(dev).manual_seed(dim + nb))
    h = LSHHasher(nb, r, dim, seed=42)
    keys = h.hash_device(x).clone()
    for spin in (2000, 0, 2000, 0):
        h.spin_wait_us = spin
        for _ in range(40):
            h.hash_device(x, out=keys)
        torch.cuda.synchronize()
        reps = 300 if dim <= 128 else 100
        t0 = time.perf_counter()
        for _ in range(reps):
            h.hash_device(x, out=keys)
        torch.cuda.synchronize()
        step = (time.perf_counter() - t0) / reps * 1e6
        h.kernel_events = []
        for _ in range(50):
            h.hash_device(x, out=keys)
        ev = h.kernel_events
        h.kernel_events = None
        s1 = sum(e[0] for e in ev) / len(ev) * 1e3
        s2 = sum(e[3] for e in ev) / len(ev) * 1e3
        print(f"{nb} x {r} x {dim}: spin_wait_us = {spin:5d}: {step:8.1f} us/step = {n / step / 1e3:6.2f} G vec/s; stage 1 {s1:7.1f} us, stage 2 {s2:6.1f} us, "
              f"rest (export, gaps, host) {step - s1 - s2:6.1f} us; step x rows/stage-1 = {s1 / step:.3f}", flush=True)
    del x, keys
