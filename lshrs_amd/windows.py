"""Windows of the signature pass: how far a fast evaluation of a projection can be from the value the host computes.

``window_coefficients`` derives the PROVEN windows - per-hyperplane coefficients, float64 arithmetic, rounded
up - from the hyperplanes alone: the products the bf16x3 split drops, the arithmetic of ``v_mfma_f32_16x16x32_bf16``
(oracle/mfma_model.c states it; tests pin it to the hardware bit for bit) and the host BLAS's own rounding, each an
elementwise bound summed by Cauchy-Schwarz.  The rest of this module is what remains of rounds 1-2: the data-independent
bounds (`bound_tau_ulps`, `bound_tau1_ulps`: ceilings for the guard's escalation) and the measured default of round 2
(`default_tau1_ulps`, statistical: tests/_adversary.py defeats it).  Pure NumPy: no GPU, no library.
"""

from __future__ import annotations

import math
from typing import Tuple

import numpy as np

__all__ = ["window_coefficients", "bound_tau_ulps", "bound_tau1_ulps", "default_tau1_ulps", "escalated_window",
           "MFMA_BF16_ERR_UNITS"]

_U = 2.0 ** -24  # unit roundoff of float32

# Error charged to ONE v_mfma_f32_16x16x32_bf16, in units of 2^-24 (|C| + sum |a_i b_i|).  The instruction is not a
# single-rounded sum: it adds its 32 products in four steps of eight, each step aligning the addends to the largest and
# truncating (tools/probes/mfma_probe.py + mfma_model*.py).  Measured on 10 686 hand-made cases - random, wide-range,
# cancelling, sticky-bit and alignment-window operands - its result is never further than 3.3 of those units from the
# exact sum (3.9 for the f16 form); tests/test_gpu_signature.py::test_mfma_bf16_step_error re-measures it.  Charged: 8.
MFMA_BF16_ERR_UNITS = 8.0


def bound_tau_ulps(dim: int) -> float:
    """Deterministic bound, in units of 2^-24 ||x|| ||p||, on |y_chain - y_BLAS| for two f32 evaluations of one
    dim-deep dot product: a single fmaf chain (the f32 kernel: gamma_dim) against the host BLAS's eight interleaved
    chains of dim/8 fmas plus a three-level tree (gamma_(dim/8+3)); sum |x_k p_k| <= ||x|| ||p||."""
    return float(dim + (dim + 7) // 8 + 3) * 1.001


def bound_tau1_ulps(dim: int) -> float:
    """Deterministic bound, same units, on |y1 - y_BLAS| for the split pass: three dropped bf16 cross terms
    (3 * 2^-16 (1 + 2^-7) sum|x p| = 774 units), one MFMA_BF16_ERR_UNITS per matrix instruction of the projection's
    accumulator (3 per 32-deep k-tile, each relative to |C| + its own products, where |C| is at most the sum of the
    absolute values of everything accumulated so far, itself at most (1 + 2^-8)^2 (1 + 2^-7) sum|x p|: (3 dim/32 + 1)
    x 1.02 of them in all) and the host BLAS's own rounding (dim/8 + 3).  The kernel widens the window by 1 % for its
    ||x|| estimate itself (taken from the bf16 high parts: >= (1 - 2^-8) ||x||)."""
    n_mfma = 3 * ((dim + 31) // 32)
    return 768.0 * (1.0 + 2.0 ** -7) + 1.02 * MFMA_BF16_ERR_UNITS * (n_mfma + 1) + float((dim + 7) // 8 + 3)


def _bf16_rne(v: np.ndarray) -> np.ndarray:
    """float32 -> nearest-even bf16, returned as float32 (finite inputs)."""
    u = np.ascontiguousarray(v, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def host_roundings(dim: int, K: int, kinds: np.ndarray) -> np.ndarray:
    """How many single roundings product k of row i passes in the host BLAS's evaluation (model 1, `lshrs_tb_model_row_dot`),
    as an ``(rows, K)`` array (0 beyond ``dim``): the vector goes block by block (4096 elements); inside a block a product of
    the 8-lane fma kernel (kind 0) is rounded by its own chain step and every later one, ``block/8 - kk//8``, then by the three
    levels of the tree; the unfused kernels round the product itself once more (kind 2: same chains; kind 1: four chains of
    ``block/4`` steps and a two-level tree); every block after the first adds one rounding to everything before it and to
    itself.  (The chains' own sums are not small, so there is no two-sided charge here.)"""
    tail = dim & 3                # (elements behind the last whole group of four: the library's scalar tail, below)
    full_dim, dim = dim, dim & ~3
    k = np.arange(K)
    block = k // 4096
    nblk = (dim + 4095) // 4096
    kk = k - 4096 * block
    blen = np.minimum(4096, dim - 4096 * block)                     # length of the block k sits in
    later = np.where(nblk > 1, nblk - np.maximum(block, 1), 0)      # additions of block sums this product is part of
    fused = (blen + 7) // 8 - kk // 8 + 3 + later              # (a block of 8 m + 4 elements: its first four go first, one step more)
    m = np.empty((len(kinds), K), dtype=np.float64)
    m[kinds == 0] = fused
    m[kinds == 2] = fused + 1
    m[kinds == 1] = blen // 4 - kk // 4 + 2 + 1 + later
    m[:, k >= dim] = 0.0
    if tail:
        # y = fma(a0, x0, y) / y + fma(a0, x0, fl(a1 x1)) / y + fma(a2, x2, fma(a0, x0, fl(a1 x1))): one more rounding for
        # everything in front, at most four for a tail product
        m[:, :dim] += 1.0
        m[:, dim:full_dim] = 4.0
    return m


def host_roundings_sdot(dim: int, K: int) -> np.ndarray:
    """`host_roundings` for a band of ONE row - the host then calls sdot, whose order is not sgemv's (`tb_model_sdot`, csrc/
    host_tiebreak.cpp): a ``(K,)`` array, 0 beyond ``dim``.  SkylakeX build (model 1): a product of 64-element step s of S is
    rounded by its own fma and every later step's, by the fold to eight lanes, by the last 32-element step if there is one, by the
    three additions of the four accumulators, the lane halves, two levels of pairs and the final rounding of (double tail +
    kernel): ``S - s + 9``; a product of the 32-element step 8; Haswell / Zen build (model 2): 32-element steps, ``S' - s + 6``.
    An element behind the last whole 32: its own product's rounding, the final one, and one for the double sums.  Both builds
    share one window (`window_coefficients` folds model 2 into 1): the elementwise maximum."""
    k = np.arange(K)
    n1, n64 = dim & ~31, dim & ~63
    m1 = np.where(k < n64, n64 // 64 - k // 64 + 9, np.where(k < n1, 8, 3))
    m2 = np.where(k < n1, n1 // 32 - k // 32 + 6, 3)
    m = np.maximum(m1, m2).astype(np.float64)
    m[k >= dim] = 0.0
    return m


def window_coefficients(planes: np.ndarray, blas_model: int, rows_per_band: int = 0
                        ) -> Tuple[np.ndarray, np.ndarray, np.ndarray, dict]:
    """The PROVEN windows of the signature pass, as per-hyperplane coefficients (float64 arithmetic, rounded up to float32):

        stage 1 of the split pass:   |y1 - y_target|      <= ||x_hi|| * coef_a[j] + ||x_mid|| * coef_b[j]
        the f32 kernel's fmaf chain: |y_chain - y_host|   <= ||x|| * coef_tie[j]

    ``x_hi = bf16(x)``, ``x_mid = bf16(x - x_hi)`` (their norms are what ``sig16_kernel`` accumulates, from the values it
    feeds the matrix cores); ``y_target`` is what decides a flagged projection: the host BLAS's value as stage 2 replays
    it (``blas_model`` 1: eight interleaved fma chains + a three-level tree for the rows the library takes four at a time,
    its unfused kernels for the ``rows_per_band % 4`` rows left over in every band - ``planes`` is the bands stacked, 0 = all
    rows of the first kind -, blocks of 4096 elements: `_hostblas.blas_order_model`, `host_roundings`) or, without a
    recognised order (``blas_model`` 0: stage 2 evaluates the f32 chain and the host engine decides the ties), the host's
    value by way of the chain (both distances: whatever stage 1 does not flag must have the HOST's sign).
    Every term is an elementwise error bound summed by Cauchy-Schwarz against a per-hyperplane constant:

    * the products the bf16x3 split drops: ``x p - (x_hi p_hi + x_hi p_mid + x_mid p_hi) = x_mid p_mid + (x_hi + x_mid) e_p
      + e_x p``, with ``|e_x,k| <= 2^-8 |x_mid,k|`` (half an ulp of the middle piece);
    * ``v_mfma_f32_16x16x32_bf16`` (oracle/mfma_model.c, bit-exact on > 1e6 probes): per STEP of eight products the result
      is within ``8 * 2^(E-24) <= 8 u max|a_k b_k|`` (seven truncated products and the accumulator) plus
      ``(1 + 2^-6) u max(|accumulator|, |result|)`` (the rounding and the adder's two cuts below the last place) of exact; the
      accumulator after step s is at most the sum of the |products| of steps <= s AND at most |y1| + those of steps > s, so
      a product in step s(k) of S is charged ``|s(k) - S/2|`` roundings: ``||p o c||`` is ~ S / (2 sqrt 3) ||p||, not S ||p||;
    * the target's own rounding: product k of the BLAS's chain j passes ``dim/8 - k//8 + 3`` single roundings (model 1, a
      row of the 8-lane kernel; `host_roundings` has the other rows), of the f32 chain ``|position(k) - K/2|`` (two-sided
      again: the chain's final value is the one under test); an unknown order: ``dim + 1``.

    Returns ``(coef_a, coef_b, coef_tie, info)``; ``info["window_units"]``: the stage-1 window of a row with
    ``||x_mid|| = 0.4 * 2^-8 ||x||`` (Gaussian-like data) in units of 2^-24 ||x|| ||p||, averaged over the hyperplanes."""
    # (hyperplanes a user assigned may hold NaN / Inf: their coefficients come out NaN / Inf - "always the exact decision" -
    #  and NumPy has nothing to warn about on the way)
    with np.errstate(invalid="ignore", over="ignore", divide="ignore", under="ignore"):
        return _window_coefficients(planes, blas_model, rows_per_band)


def _window_coefficients(planes, blas_model, rows_per_band):
    P = np.ascontiguousarray(planes, dtype=np.float32)
    num, dim = P.shape
    if blas_model == 2:           # (model 2 compiles the dim % 4 tail without contraction: at most the same number of roundings)
        blas_model = 1
    small_skx = blas_model == 3   # (the SkylakeX build's small-matrix kernels, at most eight elements: every product passes at most
    if small_skx:                 #  dim roundings - its own or its fma's, then at most dim - 1 additions)
        blas_model = 1
    K = (dim + 31) // 32 * 32
    p32 = np.zeros((num, K), dtype=np.float32)
    p32[:, :dim] = P
    ph32 = _bf16_rne(p32)
    pm32 = _bf16_rne(p32 - ph32)
    p, ph, pm = p32.astype(np.float64), ph32.astype(np.float64), pm32.astype(np.float64)
    ep = p - ph - pm
    k = np.arange(K)
    t, g = k // 32, (k % 32) // 8
    S = 12 * (K // 32)
    # Roundings a product is charged.  The accumulator after step s is the sum of the products of steps <= s - and ALSO the
    # final value y1 minus the products of steps > s: |D_s| <= min(prefix, |y1| + suffix).  Charging the steps of the first
    # half by their prefix and those of the second half by |y1| + suffix, a product of step s(k) is part of |s(k) - S/2|
    # bounds instead of S - s(k): weights that fall to zero in the middle - ||p o c|| ~ S / (2 sqrt 3) ||p||, half of what the
    # prefix alone gives.  (The |y1| it adds, (S/2) u |y1|, turns "|y1| <= W" into "|y1| <= W / (1 - S u / 2)": the slack.)
    half = S // 2

    def charged(step):
        return np.where(step < half, half - step, step - half).astype(np.float64)

    n0, n1, n2 = charged(12 * t + g), charged(12 * t + 4 + g), charged(12 * t + 8 + g)
    norm = lambda a: np.sqrt((a * a).sum(axis=1))       # noqa: E731
    u, R = _U, 1.0 + 2.0 ** -6      # per step: half an ulp (RNE) + 2^-7 ulp (second cut) + 2^-8 ulp (adder width): < 1 + 2^-6
    # the order of the f32 kernel's chain (oracle/chain_model.c): k = 32 t + 16 h + s sits at position 32 t + 2 s + h
    # (same two-sided charge: its final value y is what the tie test looks at, so |partial sum| <= min(prefix, |y| + suffix))
    pos = 32 * t + 2 * (k % 16) + (k % 32) // 16
    m_chain = np.where(pos < K // 2, K // 2 - pos, pos - K // 2 + 1).astype(np.float64)
    if blas_model == 1:
        r = int(rows_per_band)
        if r > 0:
            if num % r:
                raise ValueError("planes must be whole bands of rows_per_band hyperplanes")
            j, r4 = np.arange(num) % r, r & ~3
            kinds = np.where(j < r4, 0, np.where(((r & 3) == 1) | (j - r4 == 2), 2, 1))
        else:
            kinds = np.zeros(num, dtype=np.int64)
        m_host = host_roundings(dim, K, kinds) if r != 1 else np.broadcast_to(host_roundings_sdot(dim, K), (num, K))
        if small_skx:
            m_host = np.broadcast_to(np.where(k < dim, float(dim), 0.0), (num, K))
    else:
        m_host = np.where(k < dim, dim + 1, 0).astype(np.float64)
    # what stage 1's value is measured against is what finally DECIDES a projection it does not flag: the replayed BLAS
    # value (model 1), or - stage 2 evaluating the chain and the host engine deciding the ties - the host's value by way of
    # the chain (model 0: both distances; the stage-1 window then contains the tie window, as it must)
    a_mfma = u * R * (norm(ph * n0) + norm(pm * n1)) + 8.0 * u * (norm(ph) + norm(pm))
    b_mfma = u * R * norm(ph * n2) + 8.0 * u * norm(ph)
    a_cross = norm(ep)
    b_cross = norm(pm) + norm(ep) + 2.0 ** -8 * norm(p)
    a_tgt = u * norm(p * m_host) if blas_model == 1 else u * (norm(p * m_chain) + norm(p * m_host))
    slack = 1.0 + 1e-3 + 8.0 * K * u                    # second-order terms ((1+u)^m - 1 vs m u, errors of errors, the |y| share)
    coef_a = (a_mfma + a_cross + a_tgt) * slack
    coef_b = (b_mfma + b_cross + (1.0 + 2.0 ** -8) * a_tgt) * slack
    coef_tie = u * (norm(p * m_chain) + norm(p * m_host)) * slack
    def up(a):          # to float32, rounded up; an all-zero hyperplane keeps 0 (its y is exactly 0: never flagged)
        f = a.astype(np.float32)
        return np.where(a > 0, np.nextafter(f, np.float32(np.inf)), np.float32(0)).astype(np.float32)
    pn = norm(p)
    live = pn > 0
    unit = u * np.where(live, pn, 1.0)
    info = {"window_units": float(((coef_a + 0.4 * 2.0 ** -8 * coef_b) / unit)[live].mean()) if live.any() else 0.0,
            "window_units_worst_case_row": float(((coef_a + 2.0 ** -8 * coef_b) / unit)[live].max()) if live.any() else 0.0,
            "tie_units": float((coef_tie / unit)[live].mean()) if live.any() else 0.0,
            "terms_units": {"dropped_products": float(((a_cross + 0.4 * 2.0 ** -8 * b_cross) / unit)[live].mean()),
                            "mfma": float(((a_mfma + 0.4 * 2.0 ** -8 * b_mfma) / unit)[live].mean()),
                            "target_rounding": float((a_tgt / unit)[live].mean())} if live.any() else {}}
    return up(coef_a), up(coef_b), up(coef_tie), info


def default_tau1_ulps(dim: int) -> float:
    """The default stage-1 window: 64 units at 768-d, scaled by sqrt(768 / dim).  Stage 1's deviation from the host BLAS
    is a random walk over the dim products relative to ||x|| ||p||: measured (tools/window_by_dim.py,
    profiles/r02_window_by_dim.log) its maximum over 1e5 .. 3e5 flagged projections of Gaussian, unit-norm and
    heavy-tailed data is 15 .. 20 units x sqrt(768 / dim) at every dimension from 32 to 4096 - so this window is 3.2 .. 4.3
    times the largest deviation seen at any of them, the guard (half of it) 1.6 .. 2.1 times, and the flagged fraction
    (8.5e-5 of the projections) is the same at every dimension."""
    return 64.0 * math.sqrt(768.0 / float(dim))


def escalated_window(window: float, max_dev: float, dim: int) -> Tuple[float, str]:
    """The stage-1 window after a batch whose measured deviation came within the guard of `window`: at least twice as
    wide and at least four times the deviation seen; once that reaches the deterministic bound, the bound (and the
    mode says so).  Returns (window in units, "widened" | "bound")."""
    wider = max(2.0 * window, 4.0 * max_dev)
    bound = bound_tau1_ulps(dim)
    if wider >= bound:
        return max(bound, 4.0 * max_dev), "bound"
    return wider, "widened"
