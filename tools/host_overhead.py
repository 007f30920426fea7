import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, cProfile, pstats
from lshrs_amd import LSHHasher
h = LSHHasher(16, 16, 768, seed=42)
for n in (256, 4096, 65536):
    x = torch.randn(n, 768, device="cuda")
    keys = h.hash_device(x)
    for _ in range(200): h.hash_device(x, out=keys)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2000): h.hash_device(x, out=keys)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2000
    print(f"n={n}: {1e6*dt:.1f} us per synchronous call", flush=True)
x = torch.randn(256, 768, device="cuda"); keys = h.hash_device(x)
pr = cProfile.Profile(); pr.enable()
for _ in range(3000): h.hash_device(x, out=keys)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
