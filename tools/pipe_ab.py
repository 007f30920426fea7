#!/usr/bin/env python3
"""A/B of the host tie-break path with and without the library's chunk driver (pipeline="python": one pass, then NumPy) on the bench workload, interleaved in one process:
ms per step, stage-1 / fix-up kernel means and the driver's own timers.  usage: pipe_ab.py [rows] [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lshrs_amd import LSHHasher  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
x = torch.randn(n, 768, device=dev, generator=torch.Generator(device=dev).manual_seed(1000))
keys = torch.empty((n, 16, 2), dtype=torch.uint8, device=dev)
# (tie_replay="off": this tool compares the two drivers of the HOST tie-break path; the default hasher breaks its ties on
# the device and needs neither)
hs = {name: LSHHasher(16, 16, 768, seed=42, device=0, pipeline=name, tie_replay="off") for name in ("python", "native")}
ref = None
for name, h in hs.items():
    for _ in range(3):
        h.hash_device(x, out=keys)
    torch.cuda.synchronize()
    if ref is None:
        ref = keys.clone()
    else:
        assert torch.equal(ref, keys), "drivers disagree"
for rnd in range(rounds):
    for name, h in hs.items():
        h.kernel_events = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            h.hash_device(x, out=keys)
        torch.cuda.synchronize()
        ms = 1e2 * (time.perf_counter() - t0)
        ev, h.kernel_events = h.kernel_events, None
        if isinstance(ev[0][0], float):
            s1 = sum(e[0] for e in ev) / 10
            fx = sum(e[3] for e in ev) / 10
        else:
            s1 = sum(e[0].elapsed_time(e[3]) for e in ev) / 10
            fx = sum(e[3].elapsed_time(e[1]) for e in ev) / 10
        st = h.last_stats
        print(f"{name:7s} {ms:6.3f} ms/step = {n / ms / 1e3:6.1f} M vec/s | stage1 {s1:.3f} fix {fx:.3f} per step | "
              f"head {st.get('t_head_ms', 0):.3f} tail_count {st.get('t_tail_count_ms', 0):.3f} total {st.get('t_total_ms', 0):.3f} "
              f"wait {st.get('t_wait_ms', 0):.3f} patch {st.get('t_patch_ms', 0):.3f} (last {st.get('t_patch_last_ms', 0):.3f}) native {st.get('t_native_ms', 0):.3f}",
              flush=True)
