"""What the vendor GEMM sustains on this box, beside stage 1 of the signature pass.

Not part of the product or of bench.py: a calibration of the "matrix-core peak" the roofline fraction is quoted
against.  Runs torch.matmul (hipBLASLt / rocBLAS) for >= 2 s per shape so that the power/clock state is the steady one:

  * 8192^3 bf16: the library's best case (both operands L2/LDS-blocked at will, nothing else in the kernel)
  * (n, 768) x (768, 256) bf16: stage 1's shape, one term (stage 1 issues three of these per step and also reads X as
    f32 and splits it; the library here reads a ready-made bf16 X: half the bytes, no split)
  * the same in f32: the exact-f32 kernel's shape

and prints TFLOP/s and the fraction of the dense peak (2500 bf16 / 157.3 f32).
"""

from __future__ import annotations

import json
import sys
import time

import torch


def sustain(fn, seconds: float) -> float:
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    # size the timed run from a short probe
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / 10
    iters = max(20, int(seconds / per))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main() -> None:
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    dev = torch.device("cuda:0")
    out = []
    g = torch.Generator(dev).manual_seed(1)

    def case(name, m, n, k, dtype, peak):
        a = torch.randn(m, k, device=dev, generator=g).to(dtype)
        b = torch.randn(k, n, device=dev, generator=g).to(dtype)
        c = torch.empty(m, n, device=dev, dtype=dtype)
        s = sustain(lambda: torch.matmul(a, b, out=c), seconds)
        tf = 2.0 * m * n * k / s * 1e-12
        rec = {"case": name, "m": m, "n": n, "k": k, "dtype": str(dtype).split(".")[-1], "ms": round(s * 1e3, 4),
               "tflops": round(tf, 1), "peak": peak, "frac": round(tf / peak, 3)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del a, b, c

    case("square", 8192, 8192, 8192, torch.bfloat16, 2500.0)
    case("square-16k", 16384, 16384, 16384, torch.bfloat16, 2500.0)
    case("stage-1 shape, one term", 1_000_000, 256, 768, torch.bfloat16, 2500.0)
    case("stage-1 shape, K = 3 x 768 (three terms as one GEMM)", 1_000_000, 256, 2304, torch.bfloat16, 2500.0)
    case("f32 kernel's shape", 1_000_000, 256, 768, torch.float32, 157.3)
    case("square f32", 8192, 8192, 8192, torch.float32, 157.3)


if __name__ == "__main__":
    main()
