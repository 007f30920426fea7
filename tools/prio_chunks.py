"""Can ONE synchronous pass hide stage 2 under stage 1?  (round 5, verdict item 1)

The batch is cut into row chunks; chunk i runs on a stream of its own whose hardware queue priority FALLS with i: the
dispatcher prefers chunk 0's stage-1 workgroups while there are any, hands the CUs its last round frees to chunk 1's
stage 1, and stage 2 of chunk 0 - next on chunk 0's in-order stream - takes CUs in front of chunk 1's remaining
workgroups.  No stream waits on another: the staggering is the priorities'.  Existing ABI only (`hash_device_async`,
scratch per stream), so this measures the schedule, not new kernels.

    python tools/prio_chunks.py [seconds per figure]
"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lshrs_amd import LSHHasher

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device("cuda:0")
n, dim = 1_000_000, 768
x = torch.randn(n, dim, device=dev, generator=torch.Generator(dev).manual_seed(20240101))
h = LSHHasher(16, 16, dim, seed=42)
ref = h.hash_device(x).clone()
out = torch.empty_like(ref)

hip = ctypes.CDLL("libamdhip64.so")
lo_p, hi_p = ctypes.c_int(), ctypes.c_int()
hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo_p), ctypes.byref(hi_p))
print(f"# stream priority range: least {lo_p.value}, greatest {hi_p.value}", flush=True)


def make_stream(prio):
    s = ctypes.c_void_p()
    rc = hip.hipStreamCreateWithPriority(ctypes.byref(s), ctypes.c_uint(1), ctypes.c_int(prio))   # 1 = hipStreamNonBlocking
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


pool = {p: [make_stream(p) for _ in range(4)] for p in range(hi_p.value, lo_p.value + 1)}
ROUND = 65_536


def plan(fracs):
    """Chunk boundaries at whole rounds of workgroups (the last chunk takes the ragged end)."""
    cuts, lo = [], 0
    for f in fracs[:-1]:
        hi = min(n, max(lo + ROUND, int(round(f * n / ROUND)) * ROUND + lo))
        cuts.append((lo, hi))
        lo = hi
    cuts.append((lo, n))
    return cuts


def run(fracs, prios, steps):
    cuts = plan(fracs)
    streams = [pool[p][i] for i, p in enumerate(prios)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        hs = []
        for (lo, hi), s in zip(cuts, streams):
            with torch.cuda.stream(s):
                hs.append(h.hash_device_async(x[lo:hi], out=out[lo:hi]))
        for hd in hs:
            hd.result()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, cuts


def sync(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        h.hash_device(x, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


H, L = hi_p.value, lo_p.value
M = (H + L) // 2
cases = [
    ("sync", None, None),
    ("2 chunks 50/50 same prio", (0.5, 0.5), (M, M)),
    ("2 chunks 50/50 hi/lo", (0.5, 0.5), (H, L)),
    ("2 chunks 75/25 hi/lo", (0.75, 0.25), (H, L)),
    ("2 chunks 88/12 hi/lo", (0.88, 0.12), (H, L)),
    ("2 chunks 75/25 same", (0.75, 0.25), (M, M)),
    ("3 chunks 62/31/7 hi/mid/lo", (0.62, 0.31, 0.07), (H, M, L)),
    ("3 chunks 50/37/13 hi/mid/lo", (0.5, 0.37, 0.13), (H, M, L)),
    ("sync", None, None),
    ("2 chunks 75/25 hi/lo", (0.75, 0.25), (H, L)),
    ("3 chunks 62/31/7 hi/mid/lo", (0.62, 0.31, 0.07), (H, M, L)),
]
if H == L:
    print("# this device exposes ONE priority level: the staggering cannot come from priorities", flush=True)
for name, fracs, prios in cases:
    if fracs is None:
        sync(200)
        per = sync(50)
        steps = max(100, int(seconds / per))
        per = sync(steps)
        cuts = [(0, n)]
    else:
        if len(set(prios)) < len(prios) and len(set(prios)) > 1:
            continue
        run(fracs, prios, 200)
        per, cuts = run(fracs, prios, 50)
        steps = max(100, int(seconds / per))
        per, cuts = run(fracs, prios, steps)
    ok = torch.equal(out, ref)
    print(f"{name:32s} {steps:5d} steps  {per * 1e3:.4f} ms/step  {n / per / 1e6:7.1f} M vec/s  keys ok: {ok}  "
          f"chunks {[hi - lo for lo, hi in cuts]}", flush=True)
