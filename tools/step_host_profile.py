"""Host side of one hash_device step (1 M x 768 resident): where the interpreter spends the time the GPU waits for."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher
dev = torch.device("cuda:0")
x = torch.randn(1_000_000, 768, device=dev, generator=torch.Generator(dev).manual_seed(1))
h = LSHHasher(16, 16, 768, seed=42)
keys = h.hash_device(x).clone()
for _ in range(50):
    h.hash_device(x, out=keys)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    h.hash_device(x, out=keys)
torch.cuda.synchronize()
print(f"plain: {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(300):
    h.hash_device(x, out=keys)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
