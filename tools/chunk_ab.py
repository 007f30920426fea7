"""One synchronous `hash_device` pass as a short pipeline (ABI 6): step time by chunk plan, interleaved on one box.

    python tools/chunk_ab.py [seconds per figure] [c2|c5]

Every figure: keys == the one-launch pass, ms per step over >= `seconds`, and from HIP events riding on the dispatches the
sum of the chunks' stage-1 and stage-2 durations and the span from the first stage-1 start to the last stage-2 end.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lshrs_amd import LSHHasher

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
which = sys.argv[2] if len(sys.argv) > 2 else "c2"
dev = torch.device("cuda:0")
if which == "c5":
    n, dim, h = 5_000_000, 1536, LSHHasher(16, 32, 1536, seed=7)
    R = 32_768
else:
    n, dim, h = 1_000_000, 768, LSHHasher(16, 16, 768, seed=42)
    R = 65_536
x = torch.randn(n, dim, device=dev, generator=torch.Generator(dev).manual_seed(20240101))
h.chunking = "off"
ref = h.hash_device(x).clone()
out = torch.empty_like(ref)
rounds = n // R


def run(plan, steps):
    h.chunking = plan
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        h.hash_device(x, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def rows(*rnds):
    """Chunks of the given numbers of rounds; the last chunk takes what is left."""
    r = [k * R for k in rnds]
    return r + [n - sum(r)]


plans = [("off", "off"), ("on", "on")]
if which == "c2":
    plans += [("8|7.26", rows(8)), ("12|3.26", rows(12)), ("14|1.26", rows(14)), ("8|6|1.26", rows(8, 6)),
              ("6|5|3|1.26", rows(6, 5, 3)), ("10|4|1.26", rows(10, 4)), ("5|4|3|2|1.26", rows(5, 4, 3, 2)),
              ("8|7|0.26", rows(8, 7))]
else:
    plans += [("half", rows(rounds // 2)), ("3 x", rows(rounds // 3, rounds // 3)),
              ("6 x", rows(*([rounds // 6] * 5))), ("8 x", rows(*([rounds // 8] * 7)))]
plans += [("off", "off"), ("on", "on")]
for name, plan in plans:
    run(plan, 60 if which == "c2" else 8)
    per = run(plan, 30 if which == "c2" else 5)
    steps = max(20, int(seconds / per))
    h.kernel_events = None
    per = run(plan, steps)
    ok = torch.equal(out, ref)
    h.kernel_events = []
    run(plan, 20 if which == "c2" else 5)
    ev, h.kernel_events = h.kernel_events, None
    s1 = sum(e[0] for e in ev) / len(ev)
    s2 = sum(e[3] for e in ev) / len(ev)
    span = sum(e[6] for e in ev) / len(ev) if len(ev[0]) > 6 else float("nan")
    st = h.last_stats
    print(f"{name:14s} {per * 1e3:8.4f} ms/step {n / per / 1e6:7.1f} M vec/s  keys ok {ok}  stage1 sum {s1:.4f}  stage2 sum {s2:.4f}  "
          f"span {span:.4f}  chunks {st.get('chunks', 1)} relaunches {st['relaunches']} flagged {st.get('flagged')} "
          f"audited {st.get('audited_unflagged')}", flush=True)
