"""Socket power and clocks while short-vector hashers run back to back (as tools/power_sample.py does for the headline shape)."""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher

def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True).stdout
    keep = [l.strip() for l in out.splitlines() if re.search(r"Power|sclk|mclk", l)]
    return " | ".join(re.sub(r"\s+", " ", k) for k in keep)

dev = torch.device("cuda:0")
print("idle:", smi(), flush=True)
for nb, r, dim in ((16, 4, 128), (20, 6, 128), (16, 8, 256), (16, 16, 768)):
    x = torch.randn(1_000_000, dim, device=dev, generator=torch.Generator(dev).manual_seed(1))
    h = LSHHasher(nb, r, dim, seed=42, audit_every=0)
    keys = h.hash_device(x).clone()
    stop = False
    done = [0]
    def work():
        while not stop:
            for _ in range(50):
                h.hash_device(x, out=keys)
            done[0] += 50
    t = threading.Thread(target=work); t.start()
    time.sleep(2.0)
    d0, t0 = done[0], time.perf_counter()
    for i in range(2):
        print(f"{nb} x {r} x {dim}:", smi(), flush=True)
        time.sleep(1.0)
    rate = (done[0] - d0) / (time.perf_counter() - t0)
    stop = True; t.join()
    print(f"{nb} x {r} x {dim}: {1e6 / rate:.1f} us per step while sampled", flush=True)
    time.sleep(1.0)
