"""Is stage 1's list of flagged projections the same set from launch to launch, and does a chunk whose rows end inside a
256-row tile list anything behind its end?  (round 5: a chunked pass counted one entry more than the one-launch pass on
300 000 x 1536, 16 x 32, plan [n - 1000, 1000])

    python tools/flag_determinism.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher

dev = torch.device("cuda:0")
for nb, r, dim, n, seed in ((16, 32, 1536, 300_000, 7), (16, 16, 768, 450_000, 42)):
    h = LSHHasher(nb, r, dim, seed=seed, audit_every=0)
    x = torch.randn(n, dim, device=dev, generator=torch.Generator(dev).manual_seed(11))
    raw = torch._C._cuda_getCurrentRawStream(0)
    h.chunking = "off"
    h.stage2_sorted = False      # (the plain list is what is looked at here)
    h.hash_device(x)
    st = dict(h.last_stats)
    lst = h._replay_scratch[(0, raw)][0]
    base = torch.sort(lst[:st["flagged"]].clone())[0]
    print(nb, r, dim, n, "one launch: flagged", st["flagged"], "max row", int((base >> 21).max()), "dups", int(base.shape[0] - torch.unique(base).shape[0]))
    for plan in ([n - 1000, 1000], [n - 1000 - 24, 1000 + 24], [100_000, 100_128, n - 200_128], "on"):
        h.chunking = plan
        h.hash_device(x)
        st = dict(h.last_stats)
        lst = h._replay_scratch[(0, raw)][0]
        rows = h._chunk_rows(n)
        caps_off, lo, ents = 0, 0, []
        # (the caps of this launch: recompute as _replay_launch does)
        cap = max(int(h._flag_cap_hint), n // 4 + 4096)
        if h.tau1_ulps > 256.0:
            cap = max(cap, int(n * h.num_bands * h.rows_per_band * min(h.tau1_ulps, 4096.0) * 2.0e-6) + 4096)
        caps = [int(cap * rr // n) + 4096 for rr in rows]
        for c, (rr, cc, cnt) in enumerate(zip(rows, caps, st["chunk_flagged"])):
            e = lst[caps_off:caps_off + cnt].clone()
            bad = int(((e >> 21) >= rr).sum())
            ents.append(((e >> 21) + lo) << 21 | (e & ((1 << 21) - 1)))
            print(f"   plan {plan}: chunk {c} rows {rr} flagged {cnt} entries with row >= rows: {bad}")
            caps_off += cc
            lo += rr
        allv = torch.sort(torch.cat(ents))[0]
        a, b = set(base.tolist()), set(allv.tolist())
        print("   total", st["flagged"], "extra", [(v >> 21, v & ((1 << 21) - 1)) for v in sorted(b - a)][:6], "missing",
              [(v >> 21, v & ((1 << 21) - 1)) for v in sorted(a - b)][:6], "dups", len(allv) - len(b))
