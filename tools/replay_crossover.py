import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from lshrs_amd import LSHHasher
for dim, nb, r in ((768, 16, 16), (1536, 16, 32), (128, 16, 16)):
    ha = LSHHasher(nb, r, dim, seed=1); ha.split_min_elems = 0; ha.split_min_rows = 256     # split + replay whenever the shape allows
    hb = LSHHasher(nb, r, dim, seed=1, tie_replay="off"); hb.split_min_elems = 1 << 60      # f32 kernel + host tie-break
    for n in (512, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
        x = torch.randn(n, dim, device="cuda")
        out = torch.empty((n, nb, ha.band_bytes), dtype=torch.uint8, device="cuda")
        res = []
        for h in (ha, hb):
            for _ in range(5): h.hash_device(x, out=out)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(30): h.hash_device(x, out=out)
            torch.cuda.synchronize(); res.append((time.perf_counter() - t) / 30 * 1e6)
        print(f"dim {dim} n {n:6d}: split+replay {res[0]:7.1f} us ({ha.last_stats.get('tie_break_engine')}) | f32+host {res[1]:7.1f} us", flush=True)
