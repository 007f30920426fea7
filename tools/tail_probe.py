#!/usr/bin/env python3
"""Is the quarter-full last round of stage-1 workgroups (1 M rows = 15.26 rounds of 256 workgroups) worth anything at the power cap?
Step time and stage-1 kernel time per row for batches of whole rounds (983 040 = 15 rounds, 1 048 576 = 16) against the 1 M-row
batch.   python tools/tail_probe.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lshrs_amd import LSHHasher


def main():
    dim = 768
    dev = torch.device("cuda", 0)
    xs = {n: torch.randn(n, dim, device=dev, generator=torch.Generator(device=dev).manual_seed(n)) for n in (983_040, 1_000_000, 1_048_576)}
    h = LSHHasher(16, 16, dim, seed=42)
    for rep in range(3):
        for n, x in xs.items():
            keys = h.hash_device(x)
            for _ in range(40):
                h.hash_device(x, out=keys)
            h.kernel_events = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 60
            for _ in range(reps):
                h.hash_device(x, out=keys)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            ev, h.kernel_events = h.kernel_events, None
            s1 = sum(e[0] for e in ev) / len(ev)
            print(json.dumps({"rep": rep, "rows": n, "rounds_of_256_workgroups": round(n / 256 / 256, 3), "ms_per_step": round(1e3 * dt, 4),
                              "stage1_ms": round(s1, 4), "stage1_ns_per_row": round(1e6 * s1 / n, 4), "M_vec_per_s": round(n / dt / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
