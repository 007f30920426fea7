"""Rows built to defeat a stage-1 window that is narrower than the split pass's dropped terms.

For one hyperplane ``p`` the row ``x`` is chosen so that
  * the part of every ``x_k`` below its two leading bf16 pieces (``e_k = x_k - bf16(x_k) - bf16(x_k - bf16(x_k))``) is as
    large as that part can be (63/64 of half an ulp of the middle piece) and has the SIGN OF ``p_k`` - so the term the
    bf16x3 pass drops, ``sum e_k p_k``, is ~ +2^-17 sum 2^e_k |p_k| = 100 .. 125 units of 2^-24 ||x|| ||p|| instead of the
    few units a random row gives;
  * ``|x_k|`` is a power of two times (1 + small) with the power following ``|p_k|`` (``sum |x_k p_k|`` ~ ``||x|| ||p||``);
  * the signs of ``x_k`` are balanced and a few middle mantissa bits tuned until the exact projection is ``+target`` units:
    the reference's bit is 1, while stage 1's value sits ~100 units below zero - outside a 64-unit window, sign wrong.
Pure NumPy; used by tests/test_gpu_signature.py and tools/adversary_probe.py.
"""
from __future__ import annotations

import numpy as np

_POS = 0x403F     # low 16 bits: hi piece rounds down, middle piece rounds down, remainder +63 f32 ulps (magnitude)
_NEG = 0x4041     # ... middle piece rounds up, remainder -63 f32 ulps (magnitude)


def bf16_round(v: np.ndarray) -> np.ndarray:
    u = np.asarray(v, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split_residual(x: np.ndarray) -> np.ndarray:
    """e = x - x_h - x_m of the bf16x3 split (float64, exact)."""
    x = np.asarray(x, dtype=np.float32)
    xh = bf16_round(x)
    r = (x.astype(np.float64) - xh.astype(np.float64)).astype(np.float32)      # exact in f32
    xm = bf16_round(r)
    return x.astype(np.float64) - xh.astype(np.float64) - xm.astype(np.float64)


def adversarial_row(p: np.ndarray, target_units: float = 20.0, seed: int = 0) -> np.ndarray:
    """A float32 row for hyperplane ``p`` (float32, (dim,)) as described in the module docstring."""
    p = np.asarray(p, dtype=np.float32)
    dim = p.shape[0]
    rng = np.random.default_rng(seed)
    p64 = p.astype(np.float64)
    expo = np.clip(np.round(np.log2(np.maximum(np.abs(p64), 2.0 ** -20))), -12, 12).astype(np.int64)
    mag_bits = ((expo + 127).astype(np.uint32) << 23)                        # 2^expo, mantissa filled in below
    # balance the signs: largest terms first, each against the running sum
    order = np.argsort(-(2.0 ** expo) * np.abs(p64))
    sgn = np.ones(dim)
    run = 0.0
    for k in order:
        t = (2.0 ** expo[k]) * p64[k]
        s = -1.0 if run * t > 0 else 1.0
        sgn[k] = s
        run += s * t

    def build(mid):
        same = (sgn * np.sign(p64)) >= 0                                      # sign(x_k) == sign(p_k): remainder along the magnitude
        low = np.where(same, _POS, _NEG).astype(np.uint32) | (mid.astype(np.uint32) << 7)
        x = (mag_bits | low).view(np.float32).copy()
        return np.where(sgn < 0, -x, x).astype(np.float32)

    mid = rng.integers(0, 128, dim)
    x = build(mid)
    unit = 2.0 ** -24 * float(np.linalg.norm(x.astype(np.float64)) * np.linalg.norm(p64))
    want = target_units * unit
    # tune the free middle bits (steps of 2^(e_k - 16) |p_k|), coarse to fine, until the exact projection is the target
    steps = (2.0 ** (expo - 16.0)) * p64 * sgn                                # change of y per +1 of mid[k]
    for k in np.argsort(-np.abs(steps)):
        y = float(x.astype(np.float64) @ p64)
        delta = int(np.round((want - y) / steps[k]))
        new = int(np.clip(mid[k] + delta, 0, 127))
        if new != mid[k]:
            mid[k] = new
            x = build(mid)
        if abs(float(x.astype(np.float64) @ p64) - want) < 0.25 * unit:
            break
    return x


def tent_row(p: np.ndarray, target_units: float = 5.0, seed: int = 0) -> np.ndarray:
    """A row whose partial sums against ``p`` climb over the first half of k (every product positive) and come back down
    over the second (every product negative) to ``target_units`` of 2^-24 ||x|| ||p||: the accumulator is as large as a
    near-zero projection allows for as many steps as possible - the case the two-sided charge of the matrix instruction's
    roundings (lshrs_amd.windows.window_coefficients) is about."""
    p = np.asarray(p, dtype=np.float32)
    dim = p.shape[0]
    rng = np.random.default_rng(seed)
    p64 = p.astype(np.float64)
    mag = np.abs(rng.standard_normal(dim)) + 0.25
    sgn = np.where(np.arange(dim) < dim // 2, 1.0, -1.0) * np.where(p64 >= 0, 1.0, -1.0)
    up = float((mag[:dim // 2] * np.abs(p64[:dim // 2])).sum())
    down = float((mag[dim // 2:] * np.abs(p64[dim // 2:])).sum())
    mag[dim // 2:] *= up / down
    x = (sgn * mag).astype(np.float32)
    unit = 2.0 ** -24 * float(np.linalg.norm(x.astype(np.float64)) * np.linalg.norm(p64))
    order = np.argsort(-np.abs(p64))                         # tune the elements with the largest |p_k|: coarse to fine
    for k in order[:64]:
        y = float(x.astype(np.float64) @ p64)
        if abs(y - target_units * unit) < 0.5 * unit:
            break
        new = np.float32(x[k] - (y - target_units * unit) / p64[k])
        if new != 0 and np.sign(new) == np.sign(x[k]):
            x[k] = new
    return x


def describe(x: np.ndarray, p: np.ndarray) -> dict:
    x64, p64 = np.asarray(x, dtype=np.float64), np.asarray(p, dtype=np.float64)
    unit = 2.0 ** -24 * float(np.linalg.norm(x64) * np.linalg.norm(p64))
    e = split_residual(x)
    return {"y_units": float(x64 @ p64) / unit, "dropped_ex_p_units": float(e @ p64) / unit,
            "sum_abs_over_norms": float(np.abs(x64 * p64).sum() / (np.linalg.norm(x64) * np.linalg.norm(p64)))}
