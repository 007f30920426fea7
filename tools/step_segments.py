"""Where the host's share of a synchronous short-vector step goes: launch (Python + the C call's three dispatches), wait, finish,
and what `hash_device` adds around them (route, window, checks)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lshrs_amd import LSHHasher, _native
dev = torch.device("cuda:0")
n, nb, r, dim = 1_000_000, 16, 4, 128
x = torch.randn(n, dim, device=dev, generator=torch.Generator(dev).manual_seed(dim + nb))
h = LSHHasher(nb, r, dim, seed=42)
keys = h.hash_device(x).clone()
for _ in range(50):
    h.hash_device(x, out=keys)
ws = h._workspace(dev)
model = h._replay_model()
tau = h._tau_arg()
now = time.perf_counter_ns
seg = [0, 0, 0, 0]
reps = 400
lib = _native.load()
for _ in range(reps):
    stats = {"n": n, "tie_entries": 0, "tie_pairs": 0, "relaunches": 0}
    t0 = now()
    state = h._replay_launch(x, keys, None, ws, tau, model)
    t1 = now()
    state[0].synchronize()
    t2 = now()
    h._replay_finish(state, stats)
    t3 = now()
    seg[0] += t1 - t0; seg[1] += t2 - t1; seg[2] += t3 - t2
torch.cuda.synchronize()
t0 = now()
for _ in range(reps):
    h.hash_device(x, out=keys)
torch.cuda.synchronize()
total = (now() - t0) / reps / 1e3
print(f"launch {seg[0] / reps / 1e3:.1f} us, wait {seg[1] / reps / 1e3:.1f} us, finish {seg[2] / reps / 1e3:.1f} us; sum {sum(seg) / reps / 1e3:.1f}; hash_device step {total:.1f} us")
# the C call alone (same arguments, no Python bookkeeping): time of the ctypes call
import ctypes
scratch = h._replay_scratch[(0, torch._C._cuda_getCurrentRawStream(0))]
ptrs = scratch[9]
tc = 0
for i in range(reps):
    ptrs[5][0].done_epoch = 1000 + i
    t0 = now()
    rc = lib.lshrs_sig_hash_batch_split_replay_f32(x.data_ptr(), n, x.stride(0), ws.data_ptr(), nb, r, dim, keys.data_ptr(), ptrs[1], tau, None,
                                                   ptrs[0], ptrs[2], ptrs[3], h._tau1_arg(), model, ptrs[4][0], None, ptrs[6][0], torch._C._cuda_getCurrentRawStream(0))
    t1 = now()
    lib.lshrs_wait_done(ptrs[9][0], 1000 + i, 2_000_000, torch._C._cuda_getCurrentRawStream(0))
    t2 = now()
    tc += t1 - t0
    seg[3] += t2 - t1
print(f"bare C call {tc / reps / 1e3:.1f} us, wait behind it {seg[3] / reps / 1e3:.1f} us -> {(tc + seg[3]) / reps / 1e3:.1f} us per bare step")
