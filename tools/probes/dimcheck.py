"""16 x 16 hashers over vector lengths with odd and even k-tile counts: does hash_device return, and are the keys the oracle's?
    python tools/probes/dimcheck.py [path of a tree with lshrs_amd/]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, sys.argv[1] if len(sys.argv) > 1 else ROOT)
import numpy as np
import torch
from lshrs_amd import LSHHasher

for dim in (136, 160, 192, 224, 256, 288, 320, 352):
    h = LSHHasher(16, 16, dim, seed=42)
    x = torch.randn(200_000, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(dim))
    try:
        k = h.hash_device(x)
        st = dict(h.last_stats)
        print(dim, "ok", st.get("route"), "ratio", st.get("audit_max_window_ratio"), "flagged", st.get("flagged"), flush=True)
    except Exception as e:
        print(dim, "FAIL", str(e)[:200], flush=True)
