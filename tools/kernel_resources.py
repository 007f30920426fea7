#!/usr/bin/env python3
"""Register / LDS / spill table of the signature kernels as hipcc builds them for gfx950 (no GPU needed):
    python3 tools/kernel_resources.py [name filter ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lshrs_amd._native import CSRC, SOURCES      # noqa: E402  (one translation unit per kernel family)

out = ""
for src in SOURCES:
    out += subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", src, "-o", "/dev/null",
                           "-Rpass-analysis=kernel-resource-usage"] + [a for a in sys.argv[1:] if a.startswith("-D")],
                          capture_output=True, text=True).stderr
want = [a for a in sys.argv[1:] if not a.startswith("-D")] or ["sig16", "sig_fix", "sig_small", "sig_kernel<8, true"]
cur, rows = None, {}
for ln in out.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = cur.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z][\w \[\]/]*?): (\d+)", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':44s} VGPR AGPR spillV spillS scratch occ   LDS")
for k, v in rows.items():
    if any(w in k for w in want):
        print(f"{k:44s} {v.get('VGPRs', 0):4d} {v.get('AGPRs', 0):4d} {v.get('VGPRs Spill', 0):6d} {v.get('SGPRs Spill', 0):6d} "
              f"{v.get('ScratchSize [bytes/lane]', 0):7d} {v.get('Occupancy [waves/SIMD]', 0):3d} {v.get('LDS Size [bytes/block]', 0):6d}")
