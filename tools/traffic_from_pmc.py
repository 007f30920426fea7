#!/usr/bin/env python3
"""HBM bytes per unit from the PMC passes (tools/pmc_summary.py --json), as MI355X_MICROARCH.md prescribes: FETCH_SIZE KiB x 2
(the gfx950 correction, checked on the device copy of known size in the same passes) + WRITE_SIZE KiB.
usage: traffic_from_pmc.py summary.json > profiles/rNN_traffic.json"""
import json
import sys

t = json.load(open(sys.argv[1]))


def pick(name, grid):
    for k, v in t.items():
        kn, g, _ = k.split("|")
        if kn.startswith(name) and int(g) == grid:
            return v
    return None


out = {"note": "HBM bytes per unit from this round's rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate passes): "
               "FETCH_SIZE KiB x 2 (gfx950 correction) + WRITE_SIZE KiB; calibration: the device copy below"}
cp = [v for k, v in t.items() if "copyBuffer" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
if cp:
    big = max(cp, key=lambda v: v["FETCH_SIZE"])           # (the runtime's small internal copies share the kernel name)
    out["calibration_copy"] = {"dispatches": big["dispatches"], "fetch_KiB_counter": big["FETCH_SIZE"],
                               "write_KiB_counter": big["WRITE_SIZE"],
                               "expected_KiB_each_way": (3 * 3.0e6 + 2 * 1.5e7) / 5,
                               "note": "pmc_target.py copies 1M x 768 f32 (3.0e6 KiB) three times and 2.5M x 1536 f32 (1.5e7 KiB) "
                                       "twice: the mean is 7.8e6 KiB each way - WRITE_SIZE reads it exactly, FETCH_SIZE half of it"}
for label, name, grid, rows, alg, unit in (
        ("sig16_kernel", "sig16_kernel", (1_000_000 + 255) // 256 // 8 * 8 * 512 + (8 * 512 if ((1_000_000 + 255) // 256) % 8 else 0), 1_000_000, 3104, "row"),
        ("sig16_kernel_c5", "sig16_kernel", None, 2_500_000, 6208, "row"),
        ("cosine_kernel", "cosine_kernel", None, 10_000_000, 3084, "candidate")):
    v = pick(name, grid) if grid else None
    if v is None:       # by size: the largest (c5) / the only one
        cands = [(int(k.split("|")[1]), v2) for k, v2 in t.items() if k.split("|")[0].startswith(name) and "FETCH_SIZE" in v2]
        if label == "sig16_kernel":
            cands = [c for c in cands if c[0] < 4_000_000]
        if not cands:
            continue
        v = max(cands, key=lambda c: c[0])[1]
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    b = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
    key = "hbm_bytes_per_row" if unit == "row" else "hbm_bytes_per_candidate"
    out[label] = {("rows" if unit == "row" else "candidates"): rows, key: b / rows,
                  f"algorithmic_bytes_per_{unit}": alg, "ratio": b / rows / alg}
json.dump(out, sys.stdout, indent=1)
print()
