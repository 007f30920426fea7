#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of the bench command: the durations of the stage-1 / stage-2 dispatches of the LAST
`steps` steps in front of the parity check's launch (the timed steps), to set beside the HIP-event figures of the bench line.
usage: trace_steps.py <dir> [steps=20]"""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(path, newline="") as fh:
        rows += list(csv.DictReader(fh))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
s1 = [r for r in rows if "sig16_kernel" in r["Kernel_Name"]]
s2 = [r for r in rows if "sig_fix8_kernel" in r["Kernel_Name"]]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6   # noqa: E731
# launches of the process, in order: after_idle (W + K), settle + warm-up, the K timed steps, then the parity launch
timed1, timed2 = s1[-(steps + 1):-1], s2[-(steps + 1):-1]
out = {"sig16_launches_in_process": len(s1), "timed_steps_assumed": steps,
       "sig16_ms_mean_timed_steps": sum(map(dur, timed1)) / max(1, len(timed1)),
       "sig16_ms_mean_all": sum(map(dur, s1)) / max(1, len(s1)), "sig16_ms_min_all": min(map(dur, s1)) if s1 else None,
       "sig_fix8_ms_mean_timed_steps": sum(map(dur, timed2)) / max(1, len(timed2)),
       "first_12_sig16_ms": [round(dur(r), 4) for r in s1[:12]]}
json.dump(out, sys.stdout, indent=1)
print()
