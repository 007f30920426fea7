import time, numpy as np, torch
n = 2_097_152
d = torch.randint(0, 1 << 40, (n,), dtype=torch.int64, device="cuda")
c = torch.randint(0, 5, (1 << 20,), dtype=torch.int32, device="cuda")
def t(f, reps=5):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
ph = torch.empty(n, dtype=torch.int64, pin_memory=True); ch = torch.empty(1 << 20, dtype=torch.int32, pin_memory=True)
def d2h():
    ph.copy_(d, non_blocking=True); ch.copy_(c, non_blocking=True); torch.cuda.synchronize()
print("D2H 16 MB + 4 MB into pinned: %.2f ms" % t(d2h))
a = ph.numpy(); cc = ch.numpy()
print("copy 16 MB out of pinned: %.2f ms" % t(lambda: a.copy()))
b = a.copy()
print("copy 16 MB pageable -> pageable: %.2f ms" % t(lambda: b.copy()))
print("flatnonzero(1M int32) pinned: %.2f ms" % t(lambda: np.flatnonzero(cc)))
c2 = cc.copy()
print("flatnonzero(1M int32) pageable: %.2f ms" % t(lambda: np.flatnonzero(c2)))
live = np.flatnonzero(c2).astype(np.int64)
print("astype int64: %.2f ms" % t(lambda: np.flatnonzero(c2).astype(np.int64)))
print("counts[live] + cumsum: %.2f ms" % t(lambda: np.cumsum(c2[live], dtype=np.int64)))
print("kb view: %.2f ms" % t(lambda: np.ascontiguousarray(live.view(np.uint8).reshape(-1, 8)[:, :2])))
print("bands: %.2f ms" % t(lambda: (live >> 16).astype(np.int32)))
ids = np.arange(131072, dtype=np.int64)
print("ids distinct: %.2f ms" % t(lambda: bool((ids[1:] > ids[:-1]).all())))
print(".cpu() 16 MB: %.2f ms" % t(lambda: d.cpu()))
print("torch.empty pinned 16MB alloc (cached): %.3f ms" % t(lambda: torch.empty(n, dtype=torch.int64, pin_memory=True)))
