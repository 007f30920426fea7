#!/usr/bin/env python3
"""Timeline of one bench step from a `rocprofv3 --kernel-trace --output-format csv` run: every kernel of the LAST
step (from the last memset-free gap > 300 us backwards ... simply: the last N kernels) with its start offset,
duration and the idle gap before it.  usage: trace_gaps.py <kernel_trace.csv> [n_last_kernels]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step of the native pipeline starts with the fill of its counters: show the last COMPLETE step when there is one
fills = [i for i, r in enumerate(rows) if "fillBuffer" in r["Kernel_Name"] and (i == 0 or "fillBuffer" not in rows[i - 1]["Kernel_Name"])]
if len(fills) >= 2 and len(sys.argv) <= 2:
    rows = rows[fills[-2]:fills[-1]]
else:
    rows = rows[-n_last:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  gap {gap:7.1f} us  q={r.get('Queue_Id', '?'):>3}  {name}")
    prev_end = max(prev_end or e, e)
    busy += e - s
print(f"span {(prev_end - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us")
