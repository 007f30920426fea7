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
# GPU span of the last step's signature passes vs the step's wall time
h.kernel_events = []
torch.cuda.synchronize(); t = time.perf_counter(); h.hash_device(x, out=out); torch.cuda.synchronize(); wall = (time.perf_counter() - t) * 1e3
ev = h.kernel_events; h.kernel_events = None
span = ev[0][0].elapsed_time(ev[-1][1])
busy = sum(e[0].elapsed_time(e[1]) for e in ev)
gaps = [round(ev[i][1].elapsed_time(ev[i + 1][0]) * 1e3) for i in range(len(ev) - 1)]
print(f"one step: wall {wall:.3f} ms; first sig start -> last fix end {span:.3f} ms; sum of (sig+fix) {busy:.3f} ms; gaps between chunks (us) {gaps}; per chunk (ms) {[round(e[0].elapsed_time(e[1]), 3) for e in ev]}")
