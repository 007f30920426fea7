"""Kernel timeline of the chunked pass (ABI 6) from a rocprofv3 --kernel-trace CSV.

    rocprofv3 --kernel-trace -d <dir> -o t --output-format csv -- python3 tools/chunk_trace.py run <plan> [c2|c5]
    python3 tools/chunk_trace.py show <dir> [last_n_kernels]
plan: off | on | rounds separated by commas (8,6 = chunks of 8 and 6 rounds and the rest)
"""
import csv, glob, os, sys

if sys.argv[1] == "show":
    rows = []
    for path in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("sig16", "sig_fix", "export_counts"))]
    last = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    rows = rows[-last:]
    t0 = int(rows[0]["Start_Timestamp"])
    prev_end = 0
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:24]
        print(f"{s / 1e3:10.1f} {e / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  q {r.get('Queue_Id', '?'):>3}  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>9}  {name}")
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher

plan = sys.argv[2]
which = sys.argv[3] if len(sys.argv) > 3 else "c2"
dev = torch.device("cuda:0")
if which == "c5":
    n, dim, h, R, steps = 5_000_000, 1536, LSHHasher(16, 32, 1536, seed=7), 32_768, 12
else:
    n, dim, h, R, steps = 1_000_000, 768, LSHHasher(16, 16, 768, seed=42), 65_536, 200
x = torch.randn(n, dim, device=dev, generator=torch.Generator(dev).manual_seed(20240101))
if plan not in ("off", "on"):
    r = [int(float(k) * R) for k in plan.split(",")]
    plan = r + [n - sum(r)]
h.chunking = plan
out = torch.empty_like(h.hash_device(x))
for _ in range(steps):
    h.hash_device(x, out=out)
torch.cuda.synchronize()
