#!/usr/bin/env python3
"""bench.py — headline benchmark of the lshrs hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): vectors/sec hashed at dim=768, num_perm=256.  One *step* = one pass of
the signature path over one resident batch: every rank hashes ROWS_PER_GPU (default 1 000 000,
BASELINE config 2) synthetic N(0,1) float32 vectors already in its HBM into (rows, 16, 2) uint8
band keys in HBM, **including the tie-break that makes the keys byte-identical to the reference**
(the raw-kernel rate is reported next to it; `config.tie_break_engine` says who broke the ties: "device-replay" =
stage 2 of the split pass replaying the host BLAS's summation order, verified against this host's NumPy at first use,
or "host" = the reference's own BLAS call on the flagged pairs, overlapped chunk by chunk).  Ranks share nothing (replicated hyperplanes, no
collective in the data path): weak scaling.  Rank 0 prints ONE JSON line.

Also in the line:
  roofline      the dominant kernel of the step alone, timed with HIP events on the launch stream inside
                the timed region (start/stop events that ride on the kernel's dispatch packet, set by the
                library's pipeline driver: the kernel's own duration, every launch of every timed step).
                The default hasher takes the split-precision pass (bf16 matrix cores +
                exact f32 fix-up, same bits): its stage-1 kernel is priced against HBM (it has left the f32
                matrix roof behind; algorithmic bytes = 3 104 B per vector), with both matrix-core views
                beside it; `roofline_f32_kernel` prices the exact-f32 kernel (precision="f32") against the
                dense f32 MFMA peak as before;
  cpu_baseline  the oracle's literal restatement of the reference (per vector, per band NumPy
                calls, one thread) timed on this host on a bounded prefix of the same workload;
  rerank        BASELINE's second metric (cosine-rerank candidates/s: 1M x 768 corpus,
                10k queries x 1k candidates, config 3), N=1 only, with its HBM roofline.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DIM, NUM_PERM, BANDS, ROWS = 768, 256, 16, 16
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 matrix peak (spec); 155 measured
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
PEAK_BF16_MFMA_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 matrix peak


def pmc_traffic(kernel: str, field: str, units: float):
    """HBM bytes per launch measured by the PMC passes committed under profiles/ (None if absent)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as fh:
            return json.load(fh)[kernel][field] * units
    except (OSError, KeyError, ValueError):
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--cpu-sample-rows", type=int, default=150_000, help="rows of the workload the CPU baseline hashes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rerank", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the (untimed) parity check against the oracle")
    ap.add_argument("--async-steps", action="store_true",
                    help="hash every step through the streaming entry point hash_device_async (verified while the next "
                         "step runs) instead of hash_device (verified before it returns)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed single-launch extras (kernel_only, roofline_f32_kernel): under rocprofv3 every "
                         "signature-kernel launch of the process is then a pipeline chunk, like the timed ones")
    ap.add_argument("--backend", default=os.environ.get("LSHRS_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsals on one GPU)")
    return ap.parse_args()


def main() -> None:
    args = parse()
    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if args.gpus != world and distributed:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and not distributed:
        raise SystemExit("for --gpus > 1 launch through torch.distributed.run (one rank per GPU)")
    local_dev = local_rank % max(1, torch.cuda.device_count())   # (== local_rank on a real N-GPU node)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if distributed:
        import torch.distributed as dist

        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)

    from lshrs_amd import LSHHasher

    n = args.rows_per_gpu
    hasher = LSHHasher(BANDS, ROWS, DIM, seed=42, device=local_dev)
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    x = torch.randn(n, DIM, device=dev, generator=gen)           # resident in HBM before timing starts
    keys = torch.empty((n, BANDS, hasher.band_bytes), dtype=torch.uint8, device=dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # A step is one pass over the batch through `hash_device`, which returns verified keys.  (`--async-steps`: through
    # the streaming entry point `hash_device_async` - the verification of step i, two counters read back, runs while
    # step i + 1 is on the GPU, every step verified inside the timed region.  It hides ~25 us of host time per step and
    # gives ~17 of them back: the stage-1 kernel is power-limited and clocks lower without the pauses.)
    pending = []

    def step():
        if not args.async_steps:
            hasher.hash_device(x, out=keys)
            return
        pending.append(hasher.hash_device_async(x, out=keys))
        if len(pending) > 1:
            pending.pop(0).result()

    def drain():
        while pending:
            pending.pop(0).result()

    for _ in range(args.warmup):
        step()
    drain()
    hasher.kernel_events = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    events, hasher.kernel_events = hasher.kernel_events, None
    stats = dict(hasher.last_stats)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the step cuts its batch into chunks (one signature pass each): time every launch, weight by its rows.
    # Split-precision pass: start..mid = stage 1 (the dominant kernel), mid..end = sig_fix_kernel.
    # An entry is (start, end, rows, mid) HIP events recorded by the Python driver, or (stage-1 ms, None, rows, fix-up
    # ms) measured with HIP events on the launch stream by the library's own driver (csrc/pipeline.hip).
    split = bool(events) and all(e[3] is not None for e in events)
    native = bool(events) and isinstance(events[0][0], float)
    if native:
        kernel_ms = [e[0] for e in events]
        fix_ms = [e[3] for e in events] if split else []
    else:
        kernel_ms = [(e[0].elapsed_time(e[3]) if split else e[0].elapsed_time(e[1])) for e in events]
        fix_ms = [e[3].elapsed_time(e[1]) for e in events] if split else []
    kernel_rows = [e[2] for e in events]
    kernel_ms_total = sum(kernel_ms)
    kernel_ms_mean = kernel_ms_total / max(1, len(kernel_ms))
    rows_per_launch_mean = sum(kernel_rows) / max(1, len(kernel_rows))

    # raw-kernel-only pass of the same workload (not the headline value; reported beside it), and the exact-f32
    # kernel of precision="f32" on the same batch (one launch) for its own roofline
    raw_ms = f32_ms = None
    if rank == 0 and not args.no_extras:
        def median_ms(h, reps=5):
            h.pipeline_chunk_rows = 10**9
            ev = []
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                h.hash_device(x, out=keys, tie_break="none")
                b.record()
                ev.append((a, b))
            torch.cuda.synchronize(dev)
            return sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]

        chunk = hasher.pipeline_chunk_rows
        raw_ms = median_ms(hasher)
        hasher.pipeline_chunk_rows = chunk
        h32 = LSHHasher(BANDS, ROWS, DIM, seed=42, device=local_dev, precision="f32")
        h32.hash_device(x, out=keys, tie_break="none")
        f32_ms = median_ms(h32)

    result = None
    if rank == 0:
        total_rows = n * world
        flops_per_launch = 2.0 * DIM * NUM_PERM * rows_per_launch_mean      # 393 216 FLOP per vector
        bytes_per_launch = (4.0 * DIM + NUM_PERM / 8) * rows_per_launch_mean    # 3 104 B per vector
        achieved_tflops = flops_per_launch / (kernel_ms_mean * 1e-3) / 1e12
        achieved_gbs = bytes_per_launch / (kernel_ms_mean * 1e-3) / 1e9
        common = {
            "kernel_ms_mean": kernel_ms_mean,
            "launches_timed": len(kernel_ms),
            "launches_per_step": len(kernel_ms) / max(1, args.steps),
            "rows_per_launch_mean": rows_per_launch_mean,
            "kernel_ms_per_step": kernel_ms_total / max(1, args.steps),
            "flops_per_launch": flops_per_launch,
            "algorithmic_bytes_per_launch": bytes_per_launch,
        }
        if split:
            roofline = {
                "kernel": "sig16_kernel<RT=2, W=8>  (stage 1 of the split-precision pass: bf16x3 on v_mfma_f32_16x16x32_bf16)",
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": PEAK_HBM_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / PEAK_HBM_GBS,
                "traffic": pmc_traffic("sig_kernel_split", "hbm_bytes_per_row", rows_per_launch_mean),
                "traffic_note": "HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/r01_traffic.json: "
                                "FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), scaled to this launch's rows",
                "why_hbm": "the pass runs 3 bf16 MFMAs per f32 multiply-add and has left the f32 matrix roof (394 M vec/s) "
                           "behind; its floors are 0.39 ms (HBM, 8 TB/s) and 0.47 ms (3 x 393 GFLOP at the bf16 peak) per 1M rows",
                "mfma_views": {
                    "algorithmic_TFLOPs": achieved_tflops,
                    "frac_of_f32_mfma_peak_157.3": achieved_tflops / PEAK_F32_MFMA_TFLOPS,
                    "executed_bf16_TFLOPs": 3.0 * achieved_tflops,
                    "frac_of_bf16_mfma_peak_2500": 3.0 * achieved_tflops / PEAK_BF16_MFMA_TFLOPS,
                },
                "fix_kernel_ms_mean": sum(fix_ms) / max(1, len(fix_ms)),
                **common,
            }
        else:
            roofline = {
                "kernel": "sig_kernel<NT=8, ALIGNED, MODE=1 (keys+ties), W=4>",
                "bound": "mfma",
                "achieved": achieved_tflops,
                "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved_tflops / PEAK_F32_MFMA_TFLOPS,
                "traffic": pmc_traffic("sig_kernel", "hbm_bytes_per_row", rows_per_launch_mean),
                "traffic_note": "HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/r01_traffic.json: "
                                "FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), scaled to this launch's rows",
                "hbm_GBps_algorithmic": achieved_gbs,
                "hbm_frac_of_8TBps": achieved_gbs / PEAK_HBM_GBS,
                **common,
            }
        from lshrs_amd import _hostblas

        if stats.get("tie_break_engine") == "device-replay":
            eng, engine_threads = None, 0          # ties broken on the device: no host worker takes part in a step
        else:
            eng = None if hasher.tie_threads == 1 else _hostblas.engine(hasher.tie_threads)
            engine_threads = eng.threads if eng is not None else 1
        result = {
            "metric": "vectors/sec hashed (768-d, num_perm=256)",
            "value": total_rows * args.steps / elapsed,
            "unit": "vectors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE config 2 per GPU: 1M x 768-d f32 N(0,1) vectors, num_perm=256 (16 bands x 16 rows), "
                            "HBM-resident in and out, keys byte-identical to the reference (tie-break included)",
                "rows_per_gpu": n, "dim": DIM, "num_perm": NUM_PERM, "num_bands": BANDS, "rows_per_band": ROWS,
                "total_rows": total_rows, "sharding": f"row-sharded x{world}, replicated hyperplanes, no collective",
                "tie_break": hasher.tie_break, "tie_break_engine": stats.get("tie_break_engine", "host"),
                "tau_ulps": hasher.tau_ulps,
                "precision": hasher.precision, "tau1_ulps": hasher.tau1_ulps, "pipeline_chunk_rows": hasher.pipeline_chunk_rows,
                "pipeline_driver": stats.get("pipeline", "python"),
                "step_entry_point": "hash_device" if not args.async_steps else "hash_device_async (each step verified while the next one runs; all verified inside the timed region)",
            },
            "roofline": roofline,
            "roofline_f32_kernel": None if f32_ms is None else {
                "kernel": "sig_kernel<NT=8, ALIGNED, MODE=0, W=4> (precision='f32', one launch of the whole batch)",
                "bound": "mfma", "achieved": 2.0 * DIM * NUM_PERM * n / (f32_ms * 1e-3) / 1e12,
                "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": 2.0 * DIM * NUM_PERM * n / (f32_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "kernel_ms": f32_ms, "vectors_per_s": n / (f32_ms * 1e-3),
            },
            "kernel_only": {"ms_per_launch_median": raw_ms, "vectors_per_s": n / (raw_ms * 1e-3) if raw_ms else None},
            "tie_break_stats_last_step": stats,
            "host_tie_break": {"engine_threads": engine_threads},
        }

    # ---------------- untimed parity check of what was just measured ----------------
    if rank == 0 and not args.no_check:
        from oracle.parallel import SharedVectors, hash_shared_literal_packed

        hasher.hash_device(x, out=keys)
        m = min(n, 131_072)                  # the oracle's literal loop on every host core: a second or two
        with SharedVectors(m, DIM) as sv:
            sv.array[:] = x[:m].cpu().numpy()
            want = hash_shared_literal_packed(hasher.projections, sv)
        ok = bool(np.array_equal(keys[:m].cpu().numpy(), want))
        result["parity_check"] = {"rows": m, "bit_exact_vs_oracle": ok}
        if not ok:
            raise SystemExit("bench: keys differ from the oracle — result invalid")

    # ---------------- CPU baseline (rank 0, N=1 only) ----------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.lshrs_oracle import hash_batch_literal_packed

        m = min(n, args.cpu_sample_rows)
        xs = x[:m].cpu().numpy()
        hash_batch_literal_packed(hasher.projections, xs[:2000])  # warm BLAS
        t0 = time.perf_counter()
        hash_batch_literal_packed(hasher.projections, xs)
        dt = time.perf_counter() - t0
        result["cpu_baseline"] = {
            "value": m / dt, "unit": "vectors/s", "cores": 1, "kind": "port",
            "sample": f"first {m} rows of the same 1M x 768 workload, reference-literal per-vector/per-band NumPy "
                      f"calls ({dt:.1f} s), host cpu_count={os.cpu_count()}",
        }
        result["speedup_vs_cpu_baseline"] = result["value"] / (m / dt)
        # for honesty (SURVEY 8d ii): the best a CPU does with the same arithmetic - one batched sgemm on every core
        # the process may use, then > 0 and packbits (NOT what the reference runs, and not bit-identical to it)
        pstack = np.concatenate([np.asarray(p, dtype=np.float32) for p in hasher.projections])
        (xs[:2000] @ pstack.T)
        t0 = time.perf_counter()
        bits = (xs @ pstack.T) > 0
        np.packbits(bits.reshape(m, BANDS, ROWS), axis=2, bitorder="little")
        dt2 = time.perf_counter() - t0
        try:
            from lshrs_amd._hostblas import _core_budget
            cores = _core_budget()
        except Exception:  # pragma: no cover
            cores = os.cpu_count()
        result["cpu_best_effort"] = {
            "value": m / dt2, "unit": "vectors/s", "cores": cores, "kind": "port",
            "sample": f"same {m} rows: one batched X @ P.T (OpenBLAS sgemm, all cores of the process's budget), > 0, packbits "
                      f"({dt2:.2f} s); not the reference's per-vector path and not bit-identical to it",
        }

    # ---------------- second metric: cosine rerank (config 3), N=1 only ----------------
    if rank == 0 and world == 1 and not args.no_rerank:
        result["rerank"] = bench_rerank(torch, dev, x, np, not args.no_cpu_baseline)

    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def bench_rerank(torch, dev, corpus, np, with_cpu: bool):
    """BASELINE config 3: corpus = the 1M x 768 vectors resident on the device, 10k queries x 1k candidates."""
    from lshrs_amd.similarity import cosine_scores_device, topk_desc_device

    gen = torch.Generator(device=dev).manual_seed(7)
    m = corpus.shape[0]
    q, c = 10_000, 1_000
    qrows = torch.randperm(m, device=dev, generator=gen)[:q]
    queries = corpus[qrows] + 0.1 * torch.randn(q, DIM, device=dev, generator=gen)
    cidx = torch.randint(0, m, (q, c), device=dev, generator=gen)

    def one():
        scores, status, qstatus = cosine_scores_device(corpus, queries, cidx)
        return topk_desc_device(scores, c)

    for _ in range(2):
        one()
    torch.cuda.synchronize(dev)
    reps = 5
    ev_total, ev_cos = [], []
    for _ in range(reps):
        a, b, e = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record()
        scores, status, qstatus = cosine_scores_device(corpus, queries, cidx)
        b.record()
        topk_desc_device(scores, c)
        e.record()
        ev_total.append((a, e))
        ev_cos.append((a, b))
    torch.cuda.synchronize(dev)
    total_ms = sorted(a.elapsed_time(b) for a, b in ev_total)[reps // 2]
    cos_ms = sorted(a.elapsed_time(b) for a, b in ev_cos)[reps // 2]
    bytes_per_launch = (4.0 * DIM + 8 + 4) * q * c
    out = {
        "metric": "cosine-rerank candidates/sec (1M x 768 corpus, 10k queries x 1k candidates, k=1000)",
        "value": q * c / (total_ms * 1e-3), "unit": "candidates/s", "ms_per_pass": total_ms,
        "roofline": {
            "kernel": "cosine_kernel<true>", "bound": "hbm", "achieved": bytes_per_launch / (cos_ms * 1e-3) / 1e9,
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": bytes_per_launch / (cos_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
            "traffic": pmc_traffic("cosine_kernel", "hbm_bytes_per_candidate", q * c), "kernel_ms": cos_ms,
            "algorithmic_bytes_per_launch": bytes_per_launch,
        },
        "topk_ms": total_ms - cos_ms,
    }
    if with_cpu:
        from oracle.lshrs_oracle import top_k_cosine

        nq = 100
        qh = queries[:nq].cpu().numpy()
        ch = [corpus[cidx[i]].cpu().numpy() for i in range(nq)]
        t0 = time.perf_counter()
        for i in range(nq):
            top_k_cosine(qh[i], ch[i], k=c)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": nq * c / dt, "unit": "candidates/s", "cores": 1, "kind": "port",
                               "sample": f"{nq} of the 10k queries x 1k candidates, reference-literal top_k_cosine ({dt:.1f} s)"}
    return out


if __name__ == "__main__":
    main()
