"""Stage 2 column by column (ABI 6, lshrs_sig_sort) against the plain stage 2, interleaved on one box.

    python tools/stage2_sorted_ab.py [c5|c2|both]
Per figure: ms per synchronous step, stage-1 / stage-2 time from HIP events riding on the dispatches (stage 2 = first sort launch
.. last stage-2 launch), keys == the plain pass.
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher

which = sys.argv[1] if len(sys.argv) > 1 else "both"
dev = torch.device("cuda:0")
for name, n, dim, nb, r, seed, steps in (("c5", 5_000_000, 1536, 16, 32, 7, 10), ("c2", 1_000_000, 768, 16, 16, 42, 150)):
    if which not in (name, "both"):
        continue
    h = LSHHasher(nb, r, dim, seed=seed)
    x = torch.empty((n, dim), dtype=torch.float32, device=dev)
    g = torch.Generator(device=dev).manual_seed(5)
    for lo in range(0, n, 500_000):
        x[lo:lo + 500_000].normal_(generator=g)
    h.stage2_sorted = False
    ref = h.hash_device(x).clone()
    out = torch.empty_like(ref)
    for mode in (False, "buckets", False, "buckets"):
        h.stage2_sorted = mode
        for _ in range(steps // 3 + 2):
            h.hash_device(x, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            h.hash_device(x, out=out)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / steps
        ok = torch.equal(out, ref)
        h.kernel_events = []
        for _ in range(max(3, steps // 5)):
            h.hash_device(x, out=out)
        ev, h.kernel_events = h.kernel_events, None
        s1 = sum(e[0] for e in ev) / len(ev)
        s2 = sum(e[3] for e in ev) / len(ev)
        st = h.last_stats
        print(f"{name} stage2={str(mode):8s} {per * 1e3:8.4f} ms/step {n / per / 1e6:7.1f} M vec/s  keys ok {ok}  stage1 {s1:.4f}  stage2 {s2:.4f}  "
              f"flagged {st.get('flagged')} flips {st.get('sign_flips')} max_dev {st.get('max_dev_units'):.1f} audited {st.get('audited_unflagged')}", flush=True)
    del x, ref, out
    torch.cuda.empty_cache()
