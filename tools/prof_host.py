import cProfile, pstats, sys, time
sys.path.insert(0, "/root/repo")
import torch
from lshrs_amd import LSHHasher
n, dim = 1_000_000, 768
x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(1000))
h = LSHHasher(16, 16, dim, seed=42)
keys = h.hash_device(x)
for _ in range(50): h.hash_device(x, out=keys)
pr = cProfile.Profile()
pr.enable()
for _ in range(500): h.hash_device(x, out=keys)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
# tiny batch: how long does the host side take when the GPU work is negligible
xs = x[:512].contiguous(); ks = h.hash_device(xs)
t0 = time.perf_counter()
for _ in range(2000): h.hash_device(xs, out=ks)
print("512-row step us:", (time.perf_counter() - t0) / 2000 * 1e6)
