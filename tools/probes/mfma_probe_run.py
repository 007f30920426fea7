#!/usr/bin/env python3
"""Run hand-made operands through v_mfma_f32_16x16x32_{f16,bf16} (tools/probes/mfma_probe.hip) and save the raw results
for offline modelling (tools/probes/mfma_model.py).  GPU box only:  python tools/probes/mfma_probe_run.py gpurun_out/mfma_probe.npz"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def to_bits(vals, kind):
    v = np.asarray(vals, dtype=np.float64)
    if kind == 0:
        h = v.astype(np.float16)
        assert np.array_equal(h.astype(np.float64), v), "test value not exact in f16"
        return h.view(np.uint16)
    f = v.astype(np.float32)
    assert np.array_equal(f.astype(np.float64), v)
    u = f.view(np.uint32)
    assert not (u & 0xFFFF).any(), "test value not exact in bf16"
    return (u >> 16).astype(np.uint16)


def quant(v, kind):
    """Round float64 values to the format (nearest even)."""
    if kind == 0:
        return np.asarray(v, dtype=np.float64).astype(np.float16).astype(np.float64)
    f = np.asarray(v, dtype=np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def build_tests(kind, rng):
    A, B, C, L = [], [], [], []

    def add(a, b, c, label):
        A.append(np.asarray(a, dtype=np.float64)); B.append(np.asarray(b, dtype=np.float64)); C.append(np.float32(c)); L.append(label)

    # F0 random
    for _ in range(4096):
        add(quant(rng.standard_normal(32), kind), quant(rng.standard_normal(32), kind), rng.standard_normal() * 4, "random")
    # F0b random, wide exponent range
    for _ in range(4096):
        ea = rng.integers(-6, 7, 32); eb = rng.integers(-6, 7, 32)
        add(quant(rng.standard_normal(32) * 2.0 ** ea, kind), quant(rng.standard_normal(32) * 2.0 ** eb, kind),
            rng.standard_normal() * 2.0 ** rng.integers(-8, 9), "random_wide")
    # F1 cancellation: +L at i, -L at j, small elsewhere
    for _ in range(2048):
        a = quant(rng.standard_normal(32) * 2.0 ** -4, kind); b = quant(rng.standard_normal(32) * 2.0 ** -4, kind)
        i, j = rng.choice(32, 2, replace=False)
        e = int(rng.integers(4, 13))
        a[i], b[i] = 2.0 ** e, 2.0 ** e
        a[j], b[j] = -(2.0 ** e), 2.0 ** e
        add(a, b, rng.standard_normal() * 2.0 ** -3 if rng.random() < 0.5 else 0.0, "cancel")
    # F2 ladder: one term 2^24 (as c, or as a product at position p), m unit products
    for m in range(1, 32):
        for where in range(-1, 32, 3):
            a = np.zeros(32); b = np.zeros(32)
            slots = [k for k in range(32) if k != where]
            for k in slots[:m]:
                a[k] = 1.0; b[k] = 1.0
            c = 0.0
            if where < 0:
                c = 2.0 ** 24
            else:
                a[where] = 2.0 ** 12; b[where] = 2.0 ** 12
            add(a, b, c, f"ladder_m{m}_w{where}")
    # F3 half-ulp / sticky: c = 1, products: 2^-24 (+/- tiny)
    for tiny_e in (-26, -28, -30, -34, -40, -48):
        for sgn in (1.0, -1.0):
            for pos in (0, 5, 31):
                a = np.zeros(32); b = np.zeros(32)
                a[pos] = 2.0 ** -12; b[pos] = 2.0 ** -12
                q = (pos + 7) % 32
                a[q] = sgn * 2.0 ** (tiny_e // 2); b[q] = 2.0 ** (tiny_e - tiny_e // 2)
                add(a, b, 1.0, f"sticky_e{tiny_e}_s{int(sgn)}_p{pos}")
    # F3b: exact tie, c = 1 + k ulp: round to even either way
    for base in (1.0, 1.0 + 2.0 ** -23):
        a = np.zeros(32); b = np.zeros(32); a[3] = 2.0 ** -12; b[3] = 2.0 ** -12
        add(a, b, base, "tie")
    # F5 subnormal f16 inputs
    if kind == 0:
        for ea, eb in ((-24, 10), (-15, 0), (-20, 5), (-24, 0), (-16, -16)):
            a = np.zeros(32); b = np.zeros(32); a[0] = 2.0 ** ea; b[0] = 2.0 ** eb
            add(a, b, 0.0, f"subnormal_{ea}_{eb}")
        a = np.zeros(32); b = np.zeros(32); a[:] = 2.0 ** -24; b[:] = 1.0
        add(a, b, 0.0, "subnormal_all")
    # F6 two-group order probe: +2^24 at 0, -2^24 at j, ones elsewhere
    for j in range(1, 32):
        a = np.ones(32); b = np.ones(32)
        a[0] = 2.0 ** 12; b[0] = 2.0 ** 12; a[j] = -(2.0 ** 12); b[j] = 2.0 ** 12
        add(a, b, 0.0, f"order_j{j}")
    # F7: c huge, products medium: is sum of products formed before meeting c?
    for m in (1, 2, 3, 4, 8, 16, 32):
        a = np.zeros(32); b = np.zeros(32); a[:m] = 1.0; b[:m] = 0.5
        add(a, b, 2.0 ** 24, f"chalf_m{m}")
        add(a, b, 2.0 ** 25, f"chalf25_m{m}")
    # F8: alignment window: c = 1, one product 2^-e for e in 20..60 (is it dropped / sticky?) plus a half-ulp product
    for e in range(20, 64, 2):
        a = np.zeros(32); b = np.zeros(32)
        a[0] = 2.0 ** -12; b[0] = 2.0 ** -12        # exactly half an ulp of 1
        if kind == 0 and e > 46:
            continue
        a[1] = 2.0 ** -(e // 2); b[1] = 2.0 ** -(e - e // 2)
        add(a, b, 1.0, f"window_e{e}")
    return (np.stack([to_bits(a, kind) for a in A]), np.stack([to_bits(b, kind) for b in B]),
            np.asarray(C, dtype=np.float32), np.asarray(L))


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/mfma_probe.npz"
    lib = ctypes.CDLL(os.path.join(HERE, "mfma_probe.so"))
    lib.mfma_probe_run.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int]
    res = {}
    for kind, name in ((0, "f16"), (1, "bf16")):
        A, B, C, L = build_tests(kind, np.random.default_rng(1234 + kind))
        D = np.zeros(len(C), dtype=np.float32)
        rc = lib.mfma_probe_run(A.ctypes.data, B.ctypes.data, C.ctypes.data, D.ctypes.data, len(C), kind)
        assert rc == 0, rc
        res.update({f"{name}_A": A, f"{name}_B": B, f"{name}_C": C, f"{name}_D": D, f"{name}_L": L})
        print(name, len(C), "tests run")
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    np.savez_compressed(out, **res)
    print("saved", out)


if __name__ == "__main__":
    main()
