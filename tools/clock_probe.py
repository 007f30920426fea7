#!/usr/bin/env python3
"""Sustained-load comparison of the signature-kernel variants: wall time per launch after DVFS has settled,
and the in-kernel shader clock (s_memtime / s_memrealtime per workgroup)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher, _native
lib = _native.load()
lib.lshrs_debug_set_clock_probe.argtypes = [ctypes.c_void_p]
n = 983040  # 15 full rounds of the 128-row workgroups
h = LSHHasher(16, 16, 768, precision="f32"); h.pipeline_chunk_rows = 10**9
x = torch.randn(n, 768, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
out = torch.empty((n, 16, 2), dtype=torch.uint8, device="cuda")
mfma_cycles = 24 * 128 * 64               # per wave-tile, NT=8, dim 768
for rep in range(2):
  for (w, pipe, m) in [(4, 0, 1), (4, 1, 1), (8, 1, 1)]:
    lib.lshrs_debug_set_sig_waves(w); lib.lshrs_debug_set_sig_pipe(pipe)
    for _ in range(150):                      # ~0.5 s of back-to-back launches: let DVFS settle on this variant
        h.hash_device(x, out=out, tie_break="none")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        h.hash_device(x, out=out, tie_break="none")
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 50
    blocks = n // (w * 32 * m)
    probe = torch.zeros(2 * blocks, dtype=torch.int64, device="cuda")
    lib.lshrs_debug_set_clock_probe(probe.data_ptr())
    h.hash_device(x, out=out, tie_break="none"); torch.cuda.synchronize()
    lib.lshrs_debug_set_clock_probe(None)
    p = probe.view(-1, 2).double().cpu()
    ghz = (p[:, 0] / p[:, 1] * 0.1).median().item()
    waves_per_simd = 2 if m == 1 else 1
    util = (waves_per_simd * m * mfma_cycles) / p[:, 0].median().item()
    print(f"(W={w}, ring={pipe}): sustained {ms:.3f} ms/launch = {n/ms/1e3:.1f} M vec/s = {2*768*256*n/ms/1e9:.1f} TFLOP/s; "
          f"in-kernel clock {ghz:.3f} GHz -> peak at that clock {1024*64*ghz/1e3:.1f} TFLOP/s; MFMA pipe busy in main loop {100*util:.1f} %")
