"""CPU tests of the arithmetic model behind the proven stage-1 window (oracle/mfma_model.c) and of the window itself
(lshrs_amd.windows.window_coefficients).  No GPU: the instruction's results come from tests/golden/g9_mfma_probe.npz -
raw outputs of v_mfma_f32_16x16x32_{bf16,f16} recorded on an MI355X by tools/probes/mfma_probe2_run.py (seeded operand
families of tools/probes/mfma_cases.py, the first 2 000 cases of each) and tools/probes/mfma_probe_run.py (hand-made cases,
operands stored).  The GPU suite repeats the comparison live on the box it runs on (tests/test_gpu_signature.py)."""

from __future__ import annotations

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "probes"))


@pytest.fixture(scope="module")
def probe():
    return np.load(os.path.join(ROOT, "tests", "golden", "g9_mfma_probe.npz"))


@pytest.mark.parametrize("kind,fmt", [(1, "bf16"), (0, "f16")])
def test_model_reproduces_the_recorded_instruction_results(probe, kind, fmt):
    import mfma_cases

    from oracle.build import mfma16_model

    total = 0
    for fam in mfma_cases.FAMILIES:
        a, b, c = mfma_cases.family(fam[0], kind)
        want = probe[f"{fmt}_{fam[0]}"]
        m = len(want)
        got = mfma16_model(kind, mfma_cases.to_bits(a[:m], kind), mfma_cases.to_bits(b[:m], kind), c[:m])
        bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
        assert bad.size == 0, (fmt, fam[0], bad[:5])
        total += m
    if kind == 1:       # the single case (of 1.56 M at the time) that showed the adder's extra bit: a sum that cancels into the binade below
        got = mfma16_model(1, probe["extra_bf16_A"], probe["extra_bf16_B"], probe["extra_bf16_C"])
        assert got.view(np.uint32)[0] == probe["extra_bf16_D"].view(np.uint32)[0]
    got = mfma16_model(kind, probe[f"hand_{fmt}_A"], probe[f"hand_{fmt}_B"], probe[f"hand_{fmt}_C"])
    want = probe[f"hand_{fmt}_D"]
    bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
    assert bad.size == 0, [str(probe[f"hand_{fmt}_L"][i]) for i in bad[:10]]
    assert total == len(mfma_cases.FAMILIES) * 2000 == 38_000 and len(want) > 700


def _rows(rng, planes, dim):
    """Gaussian, scaled, wide-range, heavy-tailed, constant-sign, hyperplane-aligned and adversarial rows."""
    from tests._adversary import adversarial_row, tent_row

    rows = [rng.standard_normal((40, dim)), rng.standard_normal((20, dim)) * 2.0 ** -9, rng.standard_normal((20, dim)) * 300.0,
            rng.standard_normal((20, dim)) * np.exp2(rng.integers(-10, 11, size=(20, dim))), rng.standard_cauchy((20, dim)),
            np.abs(rng.standard_normal((10, dim)))]
    stack = np.concatenate(planes)
    rows.append(np.abs(rng.standard_normal((10, dim))) * np.sign(stack[rng.integers(0, len(stack), 10)]))   # every product > 0
    rows.append(np.stack([adversarial_row(stack[j], t, seed=int(j)) for j, t in
                          zip(rng.integers(0, len(stack), 24), rng.choice([20.0, -20.0, 3.0, 60.0], 24))]))
    rows.append(np.stack([tent_row(stack[j], t, seed=int(j)) for j, t in
                          zip(rng.integers(0, len(stack), 16), rng.choice([5.0, -5.0, 0.5, 40.0], 16))]))
    return np.concatenate(rows).astype(np.float32)


@pytest.mark.parametrize("nb,r,dim,seed", [(16, 16, 768, 42), (4, 32, 1536, 7), (8, 16, 256, 3), (4, 16, 4096, 5),
                                           (20, 10, 768, 11), (6, 13, 256, 3), (4, 7, 8192, 13)])
def test_proven_window_contains_stage1s_distance_from_the_host(nb, r, dim, seed):
    """|y1 - y_host| <= ||x_hi|| coef_a + ||x_mid|| coef_b, with y1 from the accumulator model (what the kernel computes,
    bit for bit: the GPU suite checks that) and y_host = NumPy's `P_band @ x`, the reference's own call (lsh.py:200) - for
    the BLAS-order licence of THIS host when it has one (model 1), else for any summation order (model 0's host term).
    Bands of 10, 13 and 7 rows: the rows the library's 8-lane kernel leaves over go through its unfused kernels, more
    roundings each (`windows.host_roundings`); 8192-d: two blocks of the library.
    Also: the f32 chain's distance from the host inside ||x|| coef_tie, and how much of the window the worst row used."""
    from lshrs_amd import _hostblas
    from lshrs_amd.hasher import _bf16_rne, window_coefficients
    from oracle.build import chain_project, split_stage1_model

    rng = np.random.default_rng(seed)
    planes = [rng.standard_normal((r, dim)).astype(np.float32) for _ in range(nb)]
    if seed == 3:       # user-assigned hyperplanes: an int8 grid and a heavy tail
        planes = [(rng.integers(-127, 128, (r, dim)) / 64.0).astype(np.float32) if b % 2 else
                  (rng.standard_normal((r, dim)) * np.exp(2 * rng.standard_normal((r, dim)))).astype(np.float32) for b in range(nb)]
    stack = np.concatenate(planes)
    model = int(_hostblas.blas_order_model(np.stack(planes)))
    ca, cb, ct, info = window_coefficients(stack, model, r)
    x = _rows(rng, planes, dim)
    y1 = split_stage1_model(planes, x).astype(np.float64)
    yc = chain_project(planes, x).astype(np.float64)
    yh = np.concatenate([np.stack([p @ v for v in x]) for p in planes], axis=1).astype(np.float64)     # the reference's call
    xh = _bf16_rne(x)
    xm = _bf16_rne(x - xh)
    nh, nm = np.linalg.norm(xh.astype(np.float64), axis=1), np.linalg.norm(xm.astype(np.float64), axis=1)
    nx = np.linalg.norm(x.astype(np.float64), axis=1)
    if model == 1:
        thr = nh[:, None] * ca[None, :].astype(np.float64) + nm[:, None] * cb[None, :].astype(np.float64)
        used = np.abs(y1 - yh) / thr
        assert used.max() <= 1.0, float(used.max())
        assert used.max() > 0.001             # (a bound, not an estimate: adversarial rows on heavy-tailed planes use 0.63 of it)
        tie_used = np.abs(yc - yh) / (nx[:, None] * ct[None, :].astype(np.float64))
        assert tie_used.max() <= 1.0
    # model 0 (stage 2 evaluates the chain, the host engine decides the ties - any host BLAS): the stage-1 window holds
    # stage 1's distance from the HOST by way of the chain, and therefore contains the tie window
    ca0, cb0, ct0, info0 = window_coefficients(stack, 0)
    thr0 = nh[:, None] * ca0[None, :].astype(np.float64) + nm[:, None] * cb0[None, :].astype(np.float64)
    tie0 = nx[:, None] * ct0[None, :].astype(np.float64)
    assert ((np.abs(y1 - yc) + np.abs(yc - yh)) / thr0).max() <= 1.0
    assert (np.abs(yc - yh) / tie0).max() <= 1.0
    assert (tie0 <= thr0 * 1.001).all()
    assert info0["window_units"] > info["window_units"] * (1.0 if model == 0 else 1.3)
    if (nb, r, dim) == (16, 16, 768):
        assert 300 < window_coefficients(stack, 1)[3]["window_units"] < 380      # 339: DESIGN.md §3's table


@pytest.mark.parametrize("nb,r,dim", [(6, 10, 300), (5, 7, 100), (4, 16, 200), (3, 6, 1004)])
def test_proven_tie_window_of_vectors_that_are_not_whole_k_tiles(nb, r, dim):
    """The f32 kernel's chain against the host, `|y_chain - y_host| <= ||x|| coef_tie`, where the split pass does not go:
    300-d / 100-d (8 m + 4 elements: the library's 8-lane kernels take the first four first - one chain step more) and rows
    that end inside a k-tile; bands with left-over rows.  Rows cancelled against hyperplanes of every kind among them."""
    from lshrs_amd import _hostblas
    from lshrs_amd.windows import window_coefficients
    from oracle.build import chain_project

    rng = np.random.default_rng(dim + r)
    planes = [rng.standard_normal((r, dim)).astype(np.float32) for _ in range(nb)]
    stack = np.concatenate(planes)
    model = int(_hostblas.blas_order_model(np.stack(planes)))
    ca, cb, ct, info = window_coefficients(stack, model, r)
    x = rng.standard_normal((300, dim))
    for i in range(0, 300, 3):                                  # a third of the rows: nothing left of y but the order
        p = stack[(7 * i) % len(stack)].astype(np.float64)
        x[i] -= (x[i] @ p) / (p @ p) * p
    x = x.astype(np.float32)
    yc = chain_project(planes, x).astype(np.float64)
    yh = np.concatenate([np.stack([p @ v for v in x]) for p in planes], axis=1).astype(np.float64)
    nx = np.linalg.norm(x.astype(np.float64), axis=1)
    used = np.abs(yc - yh) / (nx[:, None] * ct[None, :].astype(np.float64))
    assert used.max() <= 1.0 and used.max() > 1e-4, float(used.max())


def test_window_coefficients_of_degenerate_hyperplanes():
    from lshrs_amd.hasher import window_coefficients

    planes = np.zeros((4, 64), dtype=np.float32)
    planes[1, 5] = 1.0
    planes[2] = 2.0 ** -30
    ca, cb, ct, info = window_coefficients(planes, 1)
    assert ca[0] == 0 and cb[0] == 0 and ct[0] == 0               # a zero hyperplane: y is exactly 0, never flagged
    assert (ca[1:3] > 0).all() and np.isfinite(ca).all() and np.isfinite(cb).all()
    assert ca.dtype == np.float32


def test_a_short_adversarial_search_stays_inside_the_window():
    """tools/window_search.py for a few hundred iterations (the long runs: profiles/r03_window_search.log): rows mutated to
    maximise |y1 - y_host| / window never get past 1 - and do get well past what random rows reach."""
    import subprocess

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "window_search.py"), "3", "128", "300"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-1500:]
    best = float(out.stdout.split("best ratio")[1].split()[0])
    assert 0.1 < best < 1.0, out.stdout
