"""Socket power and clocks while stage 1 runs back to back, while the vendor's square bf16 GEMM does, and idle: samples
`rocm-smi --showpower --showclocks --showmaxpower` (read-only) from a second thread.  The evidence behind "power-limited"."""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher

def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True).stdout
    keep = [l.strip() for l in out.splitlines() if re.search(r"Power|sclk|mclk|fclk|socclk", l)]
    return " | ".join(re.sub(r"\s+", " ", k) for k in keep)

dev = torch.device("cuda:0")
x = torch.randn(1_000_000, 768, device=dev, generator=torch.Generator(dev).manual_seed(1))
h = LSHHasher(16, 16, 768, seed=42, audit_every=0)
keys = h.hash_device(x).clone()
a = torch.randn(8192, 8192, device=dev).bfloat16(); b = torch.randn(8192, 8192, device=dev).bfloat16(); c = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
print("idle:", smi(), flush=True)
for name, fn in (("stage 1 + 2 back to back (hash_device)", lambda: h.hash_device(x, out=keys)),
                 ("hipBLASLt 8192^3 bf16", lambda: torch.matmul(a, b, out=c))):
    stop = False
    def work():
        while not stop:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
    t = threading.Thread(target=work); t.start()
    time.sleep(2.0)
    for i in range(3):
        print(name + ":", smi(), flush=True)
        time.sleep(1.0)
    stop = True; t.join()
    time.sleep(1.0)
print("idle again:", smi(), flush=True)
