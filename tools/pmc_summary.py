#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (one or more passes) per (kernel, grid): mean counter value per dispatch.
usage: pmc_summary.py [--json out.json] <dir> [<dir> ...]  -> markdown table on stdout (and the same numbers as JSON)"""
import collections
import csv
import glob
import json
import os
import sys

KEEP = ("sig_kernel", "sig16_kernel", "sig16r_kernel", "sig_fixany", "sig_fix8_kernel", "sig_small", "export_", "cosine_kernel", "topk_kernel", "copy", "bucket_", "query_")
args = sys.argv[1:]
out_json = None
if args and args[0] == "--json":
    out_json, args = args[1], args[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in args:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("kernel_name") or ""
                if not any(k in name for k in KEEP):
                    continue
                short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
                key = (short, int(float(row.get("Grid_Size") or 0)), int(float(row.get("Workgroup_Size") or 0)))
                acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
counters = sorted({c for k in acc.values() for c in k})
print("| kernel | grid threads | wg | dispatches | " + " | ".join(counters) + " |")
print("|---|---|---|---|" + "---|" * len(counters))
table = {}
for k, cs in sorted(acc.items()):
    n = max(len(v) for v in cs.values())
    means = {c: sum(cs[c]) / len(cs[c]) for c in cs}
    table[f"{k[0]}|{k[1]}|{k[2]}"] = {"dispatches": n, **means}
    print(f"| {k[0]} | {k[1]} | {k[2]} | {n} | " + " | ".join(f"{means[c]:.6g}" if c in means else "" for c in counters) + " |")
if out_json:
    with open(out_json, "w") as fh:
        json.dump(table, fh, indent=1)
