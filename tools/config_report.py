#!/usr/bin/env python3
"""Measure the BASELINE.json configurations that bench.py does not put in its one line:
config 4's per-GPU shard (1.25M x 768, 256 bits) and config 5 (5M x 1536, 512 bits), each with a CPU
bit-check of the first 50k rows against the oracle.  One JSON object per line on stdout."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lshrs_amd import LSHHasher
from oracle.lshrs_oracle import hash_batch_literal_packed

def run(name, n, nb, r, dim, seed, data_seed, precision):
    h = LSHHasher(nb, r, dim, seed=seed, precision=precision)
    x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(data_seed))
    keys = torch.empty((n, nb, h.band_bytes), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        h.hash_device(x, out=keys)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        h.hash_device(x, out=keys)
    torch.cuda.synchronize()
    e2e = (time.perf_counter() - t0) / reps
    stats = {k: v for k, v in h.last_stats.items() if not k.startswith("t_")}
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); h.hash_device(x, out=keys, tie_break="none"); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    raw = sorted(a.elapsed_time(b) for a, b in ev)[reps // 2] * 1e-3
    h.hash_device(x, out=keys)
    m = 50_000
    ok = bool(np.array_equal(keys[:m].cpu().numpy(), hash_batch_literal_packed(h.projections, x[:m].cpu().numpy())))
    flops = 2.0 * dim * nb * r * n
    bytes_ = (4.0 * dim + nb * h.band_bytes) * n
    print(json.dumps({"config": name, "precision": precision, "rows": n, "dim": dim, "num_perm": nb * r, "bands_rows": [nb, r],
                      "bit_exact_vectors_per_s": n / e2e, "ms_bit_exact": 1e3 * e2e,
                      "kernel_only_vectors_per_s": n / raw, "ms_kernel_only": 1e3 * raw,
                      "kernel_tflops": flops / raw / 1e12, "frac_of_f32_mfma_peak_157.3": flops / raw / 1e12 / 157.3,
                      "hbm_GBps_algorithmic": bytes_ / raw / 1e9, "frac_of_hbm_8TBps": bytes_ / raw / 1e9 / 8000,
                      "binding_roof": ("f32 MFMA (arithmetic intensity %.0f FLOP/B)" % (flops / bytes_)) if precision == "f32" else
                                      "split-precision pass: bf16 MFMA x3 / HBM (the f32 matrix roof no longer applies)",
                      "tie_stats": stats, "cpu_bit_check_first_50k_rows": ok}), flush=True)
    assert ok

for prec in ("bf16x3", "f32"):
    run("C4 per-GPU shard (10M x 768 over 8 GPUs)", 1_250_000, 16, 16, 768, 42, 1000, prec)
    run("C5 (5M x 1536, num_perm 512)", 5_000_000, 16, 32, 1536, 7, 5, prec)
