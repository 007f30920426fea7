import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from lshrs_amd import LSHHasher
from oracle.lshrs_oracle import hash_batch_literal_packed
dev = torch.device("cuda:0")
n = 1_000_000
for nb, r, dim in ((64, 1, 100), (16, 1, 768), (64, 1, 768), (128, 1, 128), (32, 1, 102)):
    h = LSHHasher(nb, r, dim, seed=42)
    x = torch.randn(n, dim, device=dev, generator=torch.Generator(dev).manual_seed(dim + nb))
    keys = h.hash_device(x)
    st = dict(h.last_stats)
    for _ in range(10):
        h.hash_device(x, out=keys)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        h.hash_device(x, out=keys)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    m = 3000
    want = hash_batch_literal_packed(h.projections, x[:m].cpu().numpy())
    f32 = LSHHasher(nb, r, dim, seed=42, precision="f32")
    ref = f32.hash_device(x)
    print(f"{nb} x {r} x {dim}: {n / dt / 1e9:.2f} G vec/s ({dt * 1e3:.3f} ms), route {st.get('route')}, flagged {st.get('flagged')}, audited {st.get('audited_unflagged')}, "
          f"bad {st.get('audit_sign_disagreements')}, max_dev {st.get('max_dev_units', 0):.1f}, oracle {bool(np.array_equal(keys[:m].cpu().numpy(), want))}, "
          f"== f32 route on all rows {bool(torch.equal(keys, ref))} ({f32.last_stats.get('route')})", flush=True)
