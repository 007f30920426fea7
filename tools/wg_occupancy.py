"""How much of stage 1's kernel time do its workgroups account for?  Every workgroup stamps s_memrealtime (100 MHz) before
its prologue, after its main loop and at its end (lshrs_sig_opts.clock_probe); the kernel's own duration comes from the
events on its dispatch.  sum(workgroup durations) / (256 CUs x kernel duration) = the share of CU time inside workgroups;
the rest is the partial last round (1 M rows = 15.26 rounds of 256) and the hand-over between workgroups on a CU."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lshrs_amd import LSHHasher, _native
from lshrs_amd.hasher import _U

dev = torch.device("cuda:0")
lib = _native.load()
for n in (1_000_000, 1_048_576, 1_250_000):
    x = torch.randn(n, 768, device=dev, generator=torch.Generator(dev).manual_seed(1))
    h = LSHHasher(16, 16, 768, seed=42)
    keys = h.hash_device(x).clone()
    for _ in range(60):
        h.hash_device(x, out=keys)
    wgs = (n + 255) // 256
    grid = (wgs + 7) // 8 * 8
    stamps = torch.zeros(4 * grid, dtype=torch.int64, device=dev)
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
    for rep in range(5):
        cnt.zero_(); stamps.zero_()
        _native.check(lib.lshrs_sig_hash_batch_split_replay_f32(
            x.data_ptr(), n, x.stride(0), ws.data_ptr(), 16, 16, 768, keys.data_ptr(), cnt.data_ptr(), float(h.tau_ulps * _U),
            None, fl.data_ptr(), None, cap, float(h.tau1_ulps * _U), 1, None, None, ctypes.byref(opts), cur.cuda_stream), "probe")
        torch.cuda.synchronize()
        k_ms = ev[0].elapsed_time(ev[1])
        st = stamps.cpu().numpy()
        loop = st[:2 * grid].reshape(-1, 2)[:, 1].astype(np.float64) / 100.0       # us, prologue + main loop
        whole = st[2 * grid:].reshape(-1, 2)[:, 1].astype(np.float64) / 100.0      # us, to the end of the epilogue
        live = whole > 0
        res.append((k_ms, loop[live].mean(), whole[live].mean(), whole[live].sum() / (256 * k_ms * 1e3), live.sum()))
    k_ms, lp, wh, share, cntw = np.median(np.array(res), axis=0)
    print(f"n = {n}: {int(cntw)} workgroups = {cntw / 256:.2f} rounds; kernel {k_ms:.4f} ms; workgroup {wh:.2f} us (main loop {lp:.2f});"
          f" CU time inside workgroups {share:.3f}; kernel / ceil(rounds) = {k_ms * 1e3 / np.ceil(cntw / 256):.2f} us per round", flush=True)
