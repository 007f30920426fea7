"""Is stage 1 at mid-length vectors (384-d, 512-d) at the package power cap like it is at 768-d?  (what hiding the
per-workgroup prologue / epilogue of `sig16_kernel` could buy at all.)  Socket power and clocks sampled from a second thread while
`hash_device` runs back to back, per vector length; the kernel's own rate beside them."""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher

def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True).stdout
    keep = [l.strip() for l in out.splitlines() if re.search(r"Power|sclk", l)]
    return " | ".join(re.sub(r"\s+", " ", k) for k in keep)

dev = torch.device("cuda:0")
n = 1_000_000
print("idle:", smi(), flush=True)
for dim in (300, 384, 512, 768):
    x = torch.randn(n, dim, device=dev, generator=torch.Generator(dev).manual_seed(dim))
    h = LSHHasher(16, 16, dim, seed=42, audit_every=0)
    keys = h.hash_device(x).clone()
    stop = False
    def work():
        while not stop:
            for _ in range(50):
                h.hash_device(x, out=keys)
            torch.cuda.synchronize()
    t = threading.Thread(target=work); t.start()
    time.sleep(2.0)
    for i in range(3):
        print(f"dim {dim}:", smi(), flush=True)
        time.sleep(0.7)
    stop = True; t.join()
    h.kernel_events = []
    for _ in range(30):
        h.hash_device(x, out=keys)
    ev, h.kernel_events = h.kernel_events, None
    s1 = sum(e[0] for e in ev) / len(ev)
    print(f"dim {dim}: stage 1 {s1:.3f} ms = {3 * 2 * dim * 256 * n / (s1 * 1e-3) / 2.5e15:.3f} of the bf16 peak", flush=True)
    del x, keys
    time.sleep(1.0)
