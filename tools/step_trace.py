"""A few synchronous hash_device steps of a shape under rocprofv3 --kernel-trace (run), and the timeline of the last ones (show).
    rocprofv3 --kernel-trace -d <dir> -o t --output-format csv -- python3 tools/step_trace.py run <bands> <rows> <dim> [n]
    python3 tools/step_trace.py show <dir> [kernels]"""
import csv, glob, os, sys
if sys.argv[1] == "show":
    rows = []
    for path in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(path, newline="")))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("sig16", "sig_fix", "export_counts", "fix_sort", "sig_kernel"))]
    rows = rows[-(int(sys.argv[3]) if len(sys.argv) > 3 else 12):]
    t0, prev = int(rows[0]["Start_Timestamp"]), None
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:34]
        print(f"{s / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {0.0 if prev is None else (s - prev) / 1e3:6.1f}  {name}")
        prev = e
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher
nb, r, dim = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 1_000_000
h = LSHHasher(nb, r, dim, seed=42)
x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
keys = h.hash_device(x)
for _ in range(300):
    h.hash_device(x, out=keys)
torch.cuda.synchronize()
