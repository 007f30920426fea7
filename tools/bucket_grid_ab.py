"""Stage 2 on stage 1's buckets: the grid (LSHRS_BUCKET_GRID) and the slab (LSHRS_BUCKET_SLAB) of `sig_fix8_kernel<., SAMEP>`, A/B builds
of tools/ab_build.py each in a process of its own, all on one box, twice round:

    python tools/bucket_grid_ab.py [c2|c5|both] lib_a.so lib_b.so ...      ("default" = the in-tree library)
Prints per library the stage-2 time from HIP events riding on the dispatches and the synchronous step."""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
from lshrs_amd import LSHHasher
which = sys.argv[1]
for name, n, dim, nb, r, seed, steps in (("c5", 5_000_000, 1536, 16, 32, 7, 10), ("c2", 1_000_000, 768, 16, 16, 42, 150)):
    if which not in (name, "both"):
        continue
    h = LSHHasher(nb, r, dim, seed=seed)
    x = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    for lo in range(0, n, 500_000):
        x[lo:lo + 500_000].normal_(generator=g)
    out = h.hash_device(x).clone()
    for _ in range(steps // 3 + 2):
        h.hash_device(x, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        h.hash_device(x, out=out)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / steps
    h.kernel_events = []
    for _ in range(max(3, steps // 5)):
        h.hash_device(x, out=out)
    ev, h.kernel_events = h.kernel_events, None
    print(f"{name} {per * 1e3:8.4f} ms/step  stage1 {sum(e[0] for e in ev) / len(ev):.4f}  stage2 {sum(e[3] for e in ev) / len(ev):.4f}  key bytes sum {int(out.sum(dtype=torch.int64))}", flush=True)
    del x
    torch.cuda.empty_cache()
''' % ROOT

which, libs = sys.argv[1], sys.argv[2:]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ)
        if lib != "default":
            env["LSHRS_HIP_LIBRARY"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, "-c", CHILD, which], env=env, capture_output=True, text=True)
        for line in (r.stdout + r.stderr[-400:] if r.returncode else r.stdout).splitlines():
            print(f"round {rnd} {os.path.basename(lib):24s} {line}", flush=True)
