#!/usr/bin/env python3
"""Where a wave of the resident-image kernel spends its cycles: k-loop against epilogue, per 16 RT-row tile (an A/B build stamps
s_memtime around both: tools/ab_build.py lib.so -DLSHRS_AB_RES_PROBE; LSHRS_HIP_LIBRARY=lib.so python3 tools/res_phase_probe.py)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from lshrs_amd import LSHHasher, _native

dev = torch.device("cuda:0")
lib = _native.load()
n = 1_000_000
for nb, r, dim in ((16, 4, 128), (20, 6, 128), (8, 16, 128), (16, 8, 256)):
    x = torch.randn(n, dim, device=dev, generator=torch.Generator(dev).manual_seed(dim + nb))
    h = LSHHasher(nb, r, dim, seed=42)
    keys = h.hash_device(x).clone()
    for _ in range(30):
        h.hash_device(x, out=keys)
    stamps = torch.zeros(6 * 256 * 16, dtype=torch.int64, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    cur = torch.cuda.current_stream(dev)
    for e in ev:
        e.record(cur)
    opts = _native.SigOpts(events=tuple(e.cuda_event for e in ev), clock_probe=stamps.data_ptr())
    cap = n // 4 + 4096
    fl = torch.empty(cap, dtype=torch.int64, device=dev)
    cnt = torch.zeros(_native.SIG_DEVICE_COUNTERS, dtype=torch.int32, device=dev)
    ws = h._workspace(dev)
    res = []
    for rep in range(7):
        cnt.zero_()
        stamps.zero_()
        _native.check(lib.lshrs_sig_hash_batch_split_replay_f32(
            x.data_ptr(), n, x.stride(0), ws.data_ptr(), nb, r, dim, keys.data_ptr(), cnt.data_ptr(), h._tau_arg(),
            None, fl.data_ptr(), None, cap, h._tau1_arg(), h._replay_model(), None, None, ctypes.byref(opts), cur.cuda_stream), "probe")
        torch.cuda.synchronize()
        st = stamps.cpu().numpy().reshape(-1, 6)
        st = st[st[:, 5] == 1].astype(np.float64)
        tiles = st[:, 2].sum()
        if rep == 6:        # how the waves' lifetimes spread: between workgroups (CUs) and inside one
            full = stamps.cpu().numpy().reshape(-1, 6).astype(np.float64)
            wpw = int(round(st.shape[0] / 256))
            life_w = (full[:256 * wpw, 4] / 100.0).reshape(256, wpw)
            tiles_w = full[:256 * wpw, 2].reshape(256, wpw)
            xcd = np.arange(256) % 8
            print(f"   workgroup means: min {life_w.mean(1).min():.1f} max {life_w.mean(1).max():.1f} us; spread inside a workgroup (max - min): "
                  f"mean {np.ptp(life_w, axis=1).mean():.1f} us; per XCD mean life: " + " ".join(f"{life_w[xcd == k].mean():.1f}" for k in range(8))
                  + f"; tiles per wave {tiles_w.min():.0f}-{tiles_w.max():.0f}", flush=True)
        res.append((1e3 * ev[0].elapsed_time(ev[1]), st.shape[0], st[:, 0].sum() / tiles, st[:, 1].sum() / tiles,
                    (st[:, 3] / st[:, 2]).mean(), (st[:, 3] / st[:, 4]).mean() * 0.1, (st[:, 4] / 100.0).mean(), (st[:, 4] / 100.0).max(), (st[:, 4] / 100.0).min()))
    k_us, waves, main, epi, per_tile, ghz, life, lmax, lmin = np.median(np.array(res), axis=0)
    print(f"{nb} x {r} x {dim}: kernel {k_us:.1f} us, {int(waves)} waves (life {lmin:.1f} / {life:.1f} / {lmax:.1f} us min / mean / max) at {ghz:.2f} GHz; per tile and wave: "
          f"k-loop {main:.0f} cycles, epilogue {epi:.0f}, whole period {per_tile:.0f}", flush=True)
