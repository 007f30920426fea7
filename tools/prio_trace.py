"""Timeline of a chunked pass from a rocprofv3 --kernel-trace CSV (tools/prio_chunks.py's schedule).

    rocprofv3 --kernel-trace -d gpurun_out/pt -o pt --output-format csv -- python3 tools/prio_trace.py run <fracs> <prios>
    python3 tools/prio_trace.py show gpurun_out/pt [last_n_kernels]
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
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        name = r["Kernel_Name"].split("(")[0][:28]
        print(f"{s / 1e3:10.1f} {e / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  q {r.get('Queue_Id', '?'):>3}  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>9}  {name}")
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
from lshrs_amd import LSHHasher

fracs = [float(v) for v in sys.argv[2].split(",")]
prios = [int(v) for v in sys.argv[3].split(",")]
dev = torch.device("cuda:0")
n, dim = 1_000_000, 768
x = torch.randn(n, dim, device=dev, generator=torch.Generator(dev).manual_seed(20240101))
h = LSHHasher(16, 16, dim, seed=42)
out = torch.empty_like(h.hash_device(x))
hip = ctypes.CDLL("libamdhip64.so")
streams = []
for p in prios:
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithPriority(ctypes.byref(s), ctypes.c_uint(1), ctypes.c_int(p)) == 0
    streams.append(torch.cuda.ExternalStream(s.value, device=dev))
ROUND = 65_536
cuts, lo = [], 0
for f in fracs[:-1]:
    hi = min(n, int(round(f * n / ROUND)) * ROUND + lo)
    cuts.append((lo, hi)); lo = hi
cuts.append((lo, n))
for _ in range(300):
    hs = []
    for (lo, hi), s in zip(cuts, streams):
        with torch.cuda.stream(s):
            hs.append(h.hash_device_async(x[lo:hi], out=out[lo:hi]))
    for hd in hs:
        hd.result()
torch.cuda.synchronize()
