import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher
n=1_000_000
x = torch.randn(n, 768, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
out = torch.empty((n,16,2),dtype=torch.uint8,device="cuda")
prec = sys.argv[1]
h = LSHHasher(16,16,768,seed=42, precision=prec)
if len(sys.argv) > 2: h.pipeline_chunk_rows = int(sys.argv[2])
for _ in range(3): h.hash_device(x,out=out)
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(5): h.hash_device(x,out=out)
torch.cuda.synchronize()
print(prec, h.pipeline_chunk_rows, "e2e ms", (time.perf_counter()-t)/5*1e3, h.last_stats, flush=True)
