#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (one or more passes) per kernel: mean counter value per dispatch.
usage: pmc_summary.py <dir> [<dir> ...]  -> markdown table on stdout"""
import csv, glob, os, sys, collections

KEEP = ("sig_kernel", "sig16_kernel", "sig_fix8_kernel", "export_ties", "cosine_kernel", "topk_kernel", "copy", "bucket_")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("kernel_name") or ""
                if not any(k in name for k in KEEP):
                    continue
                short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
counters = sorted({c for k in acc.values() for c in k})
print("| kernel | dispatches | " + " | ".join(counters) + " |")
print("|---|---|" + "---|" * len(counters))
for k, cs in sorted(acc.items()):
    n = max(len(v) for v in cs.values())
    print(f"| {k} | {n} | " + " | ".join(f"{sum(cs[c])/len(cs[c]):.6g}" if c in cs else "" for c in counters) + " |")
