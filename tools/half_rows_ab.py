"""Stage 1's two workgroup shapes against each other (round 6): 256-row workgroups, one per CU, and 128-row workgroups, two per
CU (sig16_kernel<., ., 4>) - same keys, same flagged lists; what differs is what overlaps.  A child process per measurement
(LSHRS_SIG16_HALF_MAX_TILES is read once), A and B alternating on one box.
    python tools/half_rows_ab.py [dims ...]          (child: --child DIM BANDS ROWS; HALF_AB_ROWS=n: rows per batch)"""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(dim: int, bands: int, rows: int) -> None:
    import torch
    from lshrs_amd import LSHHasher

    n = int(os.environ.get("HALF_AB_ROWS", "0")) or (1_000_000 if bands * rows * dim <= 256 * 1024 else 500_000)
    h = LSHHasher(bands, rows, dim, seed=42)
    x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(dim))
    keys = h.hash_device(x)
    reps = 40 if n >= 200_000 else 200
    for _ in range(reps):
        h.hash_device(x, out=keys)
    h.kernel_events = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        h.hash_device(x, out=keys)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ev, h.kernel_events = h.kernel_events, None
    s1 = sum(e[0] for e in ev) / max(1, len(ev))
    digest = hashlib.sha1(keys.cpu().numpy().tobytes()).hexdigest()[:12]
    flop = 3 * 2 * dim * bands * rows * n
    print(f"{s1:.4f} {dt * 1e3:.4f} {n / dt / 1e6:.1f} {flop / (s1 * 1e-3) / 2.5e15:.3f} {digest} {h.last_stats.get('route')} "
          f"{h.last_stats.get('flagged', 0)}")


def main() -> None:
    shapes = [(int(a), 16, 16) for a in sys.argv[1:]] or [(300, 16, 16), (384, 16, 16), (512, 16, 16), (768, 16, 16), (768, 32, 16)]
    for dim, b, r in shapes:
        for rnd in range(3):
            row = {}
            for name, val in (("256", "0"), ("128", "64")):
                env = dict(os.environ, LSHRS_SIG16_HALF_MAX_TILES=val)
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(dim), str(b), str(r)], env=env,
                                     capture_output=True, text=True, timeout=300)
                if out.returncode != 0:
                    print(out.stderr[-2000:])
                    raise SystemExit(1)
                row[name] = out.stdout.strip().split()
            a, h = row["256"], row["128"]
            same = a[4] == h[4] and a[6] == h[6]
            print(f"{b}x{r}x{dim} round {rnd}: stage1 {a[0]} -> {h[0]} ms ({float(a[0]) / float(h[0]):.3f}x), of bf16 peak {a[3]} -> {h[3]}, "
                  f"step {a[2]} -> {h[2]} M vec/s ({float(h[2]) / float(a[2]):.3f}x), keys {'equal' if same else 'DIFFER'} {a[5]}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    else:
        main()
