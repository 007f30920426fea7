#!/usr/bin/env python3
"""Build a second copy of liblshrs_hip.so with extra hipcc flags (for A/B runs: LSHRS_HIP_LIBRARY=<path> python ...).
    python tools/ab_build.py gpurun_out/lib_noslp.so -fno-slp-vectorize"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, flags = sys.argv[1], sys.argv[2:]
csrc = os.path.join(ROOT, "lshrs_amd", "csrc")
objs = []
sys.path.insert(0, ROOT)
from lshrs_amd._native import UNITS          # noqa: E402  (the library's translation units)

for src in [u + ".hip" for u in UNITS]:
    obj = out + "." + src + ".o"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                    *flags, "-c", os.path.join(csrc, src), "-o", obj], check=True)
    objs.append(obj)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", out], check=True)
for o in objs:
    os.remove(o)
print("built", out)
