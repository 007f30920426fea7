"""Host side of one hash_device step on a short shape (1 M x 128, 20 x 6): where the interpreter spends the time the GPU waits for."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher
dev = torch.device("cuda:0")
for n in (1_000_000, 512):
    x = torch.randn(n, 128, device=dev, generator=torch.Generator(dev).manual_seed(1))
    h = LSHHasher(20, 6, 128, seed=42)
    keys = h.hash_device(x).clone()
    for _ in range(50):
        h.hash_device(x, out=keys)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        h.hash_device(x, out=keys)
    torch.cuda.synchronize()
    print(f"n = {n}: {(time.perf_counter() - t0) / 500 * 1e3:.4f} ms/step", h.last_stats.get("route"))
pr = cProfile.Profile(); pr.enable()
for _ in range(2000):
    h.hash_device(x, out=keys)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
