#!/usr/bin/env python3
"""Offline: which arithmetic model reproduces the raw v_mfma_f32_16x16x32_{f16,bf16} results of mfma_probe.py bit for bit?
    python tools/probes/mfma_model.py gpurun_out/mfma_probe.npz"""
import math
import sys
from fractions import Fraction

import numpy as np


def bits_to_float(u16, kind):
    if kind == "f16":
        return u16.view(np.float16).astype(np.float64)
    return (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def round_f32(q: Fraction, mode="rne") -> float:
    """Correctly round an exact rational to float32 (subnormals kept)."""
    if q == 0:
        return 0.0
    s = -1 if q < 0 else 1
    a = abs(q)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    e = max(e, -126)
    ulp = Fraction(2) ** (e - 23)
    n = a / ulp
    fl = n.numerator // n.denominator
    rem = n - fl
    if mode == "rne":
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (fl & 1)):
            fl += 1
    elif mode == "rz":
        pass
    return s * float(fl * ulp)


def trunc_align(terms, width, final="rne", trunc="rz"):
    """Align every addend to the largest exponent, keep `width` bits below it (truncate toward zero or floor), add, round."""
    nz = [t for t in terms if t != 0]
    if not nz:
        return 0.0
    emax = max(math.floor(math.log2(abs(float(t)))) if abs(t) >= Fraction(1, 2 ** 1000) else -1000 for t in nz)
    # exact exponent
    def expo(t):
        a = abs(t)
        e = a.numerator.bit_length() - a.denominator.bit_length()
        if Fraction(2) ** e > a:
            e -= 1
        return e
    emax = max(expo(t) for t in nz)
    q = Fraction(2) ** (emax - width)
    tot = Fraction(0)
    for t in nz:
        n = t / q
        if trunc == "rz":
            k = abs(n.numerator) // n.denominator
            k = -k if n < 0 else k
        else:
            k = n.numerator // n.denominator
        tot += k * q
    return round_f32(tot, final)


def main():
    z = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/mfma_probe.npz")
    for kind in ("f16", "bf16"):
        A = bits_to_float(z[f"{kind}_A"], kind); B = bits_to_float(z[f"{kind}_B"], kind)
        C = z[f"{kind}_C"].astype(np.float64); D = z[f"{kind}_D"]; L = z[f"{kind}_L"]
        n = len(C)
        models = {}
        prods = [[Fraction(float(A[t, k])) * Fraction(float(B[t, k])) for k in range(32)] for t in range(n)]
        cs = [Fraction(float(C[t])) for t in range(n)]

        def run(name, fn):
            models[name] = np.array([fn(prods[t], cs[t]) for t in range(n)], dtype=np.float32)

        run("exact_rne", lambda p, c: round_f32(sum(p) + c, "rne"))
        run("exact_rz", lambda p, c: round_f32(sum(p) + c, "rz"))
        run("sum_then_c", lambda p, c: round_f32(Fraction(round_f32(sum(p))) + c))

        def chain(p, c, g):
            acc = c
            for i in range(0, 32, g):
                acc = Fraction(round_f32(acc + sum(p[i:i + g])))
            return float(acc)
        for g in (1, 2, 4, 8, 16):
            run(f"chain_g{g}", lambda p, c, g=g: chain(p, c, g))
        for w in (24, 25, 26, 27, 28, 30, 32):
            run(f"align_w{w}_rne", lambda p, c, w=w: trunc_align(p + [c], w, "rne"))
            run(f"align_w{w}_rz", lambda p, c, w=w: trunc_align(p + [c], w, "rz"))
        fam = np.array([s.split("_")[0] for s in L])
        print(f"==== {kind}: {n} tests")
        for name, M in models.items():
            eq = (M.view(np.uint32) == D.view(np.uint32)) | ((M == 0) & (D == 0))
            per = {f: f"{int(eq[fam == f].sum())}/{int((fam == f).sum())}" for f in sorted(set(fam))}
            print(f"{name:16s} total {int(eq.sum())}/{n}  {per}")
        # error of the hardware result against the exact sum, in units of 2^-24 (|c| + sum |p|)
        ex = np.array([float(sum(prods[t]) + cs[t]) for t in range(n)])
        mag = np.array([float(abs(cs[t]) + sum(abs(p) for p in prods[t])) for t in range(n)])
        ok = mag > 0
        err = np.abs(D.astype(np.float64)[ok] - ex[ok]) / (2.0 ** -24 * mag[ok])
        print(f"max |D - exact| / (2^-24 (|c| + sum|ab|)) = {err.max():.4f}; vs |D|: "
              f"{(np.abs(D.astype(np.float64)[ok] - ex[ok]) / np.maximum(2.0 ** -24 * np.abs(ex[ok]), 1e-300)).max():.4f}")
        for lab in ("subnormal", "sticky", "window", "tie", "chalf", "chalf25", "order", "ladder"):
            idx = np.flatnonzero(fam == lab)[:40]
            if idx.size:
                print(lab, [(str(L[i]), float(D[i]), float(models["exact_rne"][i])) for i in idx][:12])


if __name__ == "__main__":
    main()
