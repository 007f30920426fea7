#!/usr/bin/env python3
"""What the audit of un-flagged projections costs a BASELINE config-2 step (1 M x 768, proven window): the same batch through a
hasher with the audit (default: ~4096 sampled projections per launch) and one without, interleaved on one box, 12 rounds of
20 steps each, order alternating; medians.  -> profiles/r04_audit_cost.log"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lshrs_amd import LSHHasher

x = torch.randn(1_000_000, 768, device="cuda", generator=torch.Generator("cuda").manual_seed(5))
out = torch.empty((1_000_000, 16, 2), dtype=torch.uint8, device="cuda")
hs = {"audit on (4096 per launch)": LSHHasher(16, 16, 768, seed=42), "audit off": LSHHasher(16, 16, 768, seed=42, audit_unflagged=0),
      "audit on (32768 per launch)": LSHHasher(16, 16, 768, seed=42, audit_unflagged=32768)}
for h in hs.values():
    h.kernel_events = []
    for _ in range(60):
        h.hash_device(x, out=out)
times = {k: [] for k in hs}
k1 = {k: [] for k in hs}
k2 = {k: [] for k in hs}
names = list(hs)
for rnd in range(12):
    order = names if rnd % 2 == 0 else names[::-1]
    for name in order:
        h = hs[name]
        h.kernel_events.clear()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            h.hash_device(x, out=out)
        b.record()
        torch.cuda.synchronize()
        times[name].append(a.elapsed_time(b) / 20)
        k1[name] += [e[0] for e in h.kernel_events]
        k2[name] += [e[3] for e in h.kernel_events]
med = lambda v: sorted(v)[len(v) // 2]      # noqa: E731
base = med(times["audit off"])
for name in names:
    st = hs[name].last_stats
    print(f"{name:30s} step {med(times[name]):.4f} ms ({100 * (med(times[name]) / base - 1):+.2f} %)   stage 1 {med(k1[name]):.4f} ms   "
          f"stage 2 {med(k2[name]):.4f} ms   audited per launch {st.get('audited_unflagged')}, sign disagreements "
          f"{st.get('audit_sign_disagreements')}, max |y1 - y_host| / window {st.get('audit_max_window_ratio'):.3f}")
