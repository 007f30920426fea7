#!/usr/bin/env python3
"""GPU box: run the seeded families of mfma_cases.py through both instructions, save the raw results only.
python tools/probes/mfma_probe2_run.py gpurun_out/mfma_probe2.npz"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import mfma_cases as mc  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/mfma_probe2.npz"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lib = ctypes.CDLL(os.path.join(HERE, "mfma_probe.so"))
    lib.mfma_probe_run.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int]
    res = {}
    for kind, fmt in ((0, "f16"), (1, "bf16")):
        for fam in mc.FAMILIES:
            a, b, c = mc.family(fam[0], kind, seed=seed)
            A, B = np.ascontiguousarray(mc.to_bits(a, kind)), np.ascontiguousarray(mc.to_bits(b, kind))
            D = np.zeros(len(c), dtype=np.float32)
            rc = lib.mfma_probe_run(A.ctypes.data, B.ctypes.data, c.ctypes.data, D.ctypes.data, len(c), kind)
            assert rc == 0, rc
            res[f"{fmt}_{fam[0]}"] = D
        print(fmt, "done", flush=True)
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    np.savez_compressed(out, **res)
    print("saved", out)


if __name__ == "__main__":
    main()
