"""Seeded operand families for the second probe of v_mfma_f32_16x16x32_{f16,bf16}: the GPU box stores only the results
(4 bytes per case); the offline model regenerates the operands from the same seeds (same NumPy, same streams)."""
import numpy as np


def quant(v, kind):
    """Round float64 values to the format (nearest even): kind 0 = f16, 1 = bf16."""
    if kind == 0:
        return np.asarray(v, dtype=np.float64).astype(np.float16).astype(np.float64)
    f = np.asarray(v, dtype=np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def to_bits(v, kind):
    if kind == 0:
        return np.asarray(v, dtype=np.float64).astype(np.float16).view(np.uint16)
    return (np.asarray(v, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


# (name, products in use (the first m of group 0 unless "full"), exponent spread of a and b, C kind)
FAMILIES = [
    ("one_c", 1, 6, "wide"), ("two_c", 2, 6, "wide"), ("two_0", 2, 8, "zero"), ("three_c", 3, 6, "wide"),
    ("four_c", 4, 6, "wide"), ("eight_0", 8, 6, "zero"), ("eight_c", 8, 6, "wide"), ("eight_n", 8, 0, "narrow"),
    ("g2_c", 16, 6, "wide"), ("full_n", 32, 0, "narrow"), ("full_w", 32, 6, "wide"), ("full_big_c", 32, 2, "big"),
    ("full_chain", 32, 1, "chain"),
    # accumulators next to a power of two, products sized to carry the sum across it (either way): the cancellation /
    # carry cases in which the adder's width below the accumulator's last place shows
    ("cross_dn_1", 1, 0, "cross_dn"), ("cross_dn_8", 8, 2, "cross_dn"), ("cross_up_1", 1, 0, "cross_up"),
    ("cross_up_8", 8, 2, "cross_up"), ("cross_dn_full", 32, 2, "cross_dn"), ("cancel_deep", 8, 1, "cancel_deep"),
]
CASES_PER_FAMILY = 40_000


def family(name, kind, seed=0):
    """-> (A (n, 32) float64, B (n, 32) float64, C (n,) float32), values exact in the format."""
    spec = {f[0]: f for f in FAMILIES}[name]
    _, m, spread, ckind = spec
    rng = np.random.default_rng([seed, kind, [f[0] for f in FAMILIES].index(name)])
    n = CASES_PER_FAMILY
    ea = rng.integers(-spread, spread + 1, (n, 32)) if spread else np.zeros((n, 32), dtype=np.int64)
    eb = rng.integers(-spread, spread + 1, (n, 32)) if spread else np.zeros((n, 32), dtype=np.int64)
    a = quant(rng.standard_normal((n, 32)) * 2.0 ** ea, kind)
    b = quant(rng.standard_normal((n, 32)) * 2.0 ** eb, kind)
    a[:, m:] = 0.0
    b[:, m:] = 0.0
    if ckind in ("cross_dn", "cross_up", "cancel_deep"):
        e = rng.integers(-4, 13, n).astype(np.float64)
        d = rng.integers(3, 15, n).astype(np.float64)             # the products sit 2^-d below the accumulator
        sc = 2.0 ** np.floor((e - d) / 2.0)
        a = quant(a * sc[:, None], kind)
        b = quant(b * (2.0 ** (e - d) / sc)[:, None], kind)
        a[:, m:] = 0.0
        b[:, m:] = 0.0
        sgn = np.where(rng.random(n) < 0.5, -1.0, 1.0)
        if ckind == "cross_dn":                                   # just above 2^e: a negative product sum drops a binade
            c = sgn * 2.0 ** e * (1.0 + rng.random(n) * 2.0 ** -(d - 1))
        elif ckind == "cross_up":                                 # just below 2^(e+1)
            c = sgn * 2.0 ** e * (2.0 - rng.random(n) * 2.0 ** -(d - 1))
        else:                                                     # the product sum nearly cancels the accumulator
            s8 = (a * b).sum(axis=1)
            c = -s8 * (1.0 + rng.standard_normal(n) * 2.0 ** -rng.integers(2, 12, n))
        return a, b, c.astype(np.float32)
    if ckind == "zero":
        c = np.zeros(n)
    elif ckind == "wide":
        c = rng.standard_normal(n) * 2.0 ** rng.integers(-10, 11, n)
    elif ckind == "narrow":
        c = rng.standard_normal(n) * 4.0
    elif ckind == "big":          # an accumulator that dwarfs the products: what a long dot product looks like late on
        c = rng.standard_normal(n) * 2.0 ** rng.integers(4, 12, n)
    else:                         # "chain": c of the size the sum of ~24 such instructions reaches
        c = rng.standard_normal(n) * 2.0 ** rng.integers(1, 6, n)
    return a, b, c.astype(np.float32)
