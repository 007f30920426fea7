"""Band/row auto-configuration used by the ``LSHRS`` constructor.

One-off scalar maths that runs once per index (not part of the accelerated path); it is
here only so that ``LSHRS(dim=..., num_perm=...)`` picks the *same* (bands, rows) split as
the reference and therefore produces the same keys.  Behaviour follows
``get_optimal_config`` (lshrs/utils/br.py:325-395): table lookup for 2^12..2^16 hash bits,
otherwise a search over the factor pairs of ``num_perm`` minimising false-positive +
false-negative area of the S-curve, otherwise the most square factor pair.
``tests/golden/g6_autoconfig.json`` pins 40 (num_perm, threshold) points of the reference.
"""

from __future__ import annotations

import math
from typing import Optional, Tuple

__all__ = ["get_optimal_config", "lsh_threshold", "collision_probability", "false_rates"]

# (bands, rows) published by the reference for large signatures — data from br.py:38-78
_TABLE = {
    4096: {0.5: (512, 8), 0.7: (256, 16), 0.85: (128, 32), 0.9: (64, 64), 0.95: (32, 128)},
    8192: {0.4: (1024, 8), 0.7: (512, 16), 0.8: (256, 32), 0.85: (256, 32), 0.9: (128, 64), 0.95: (64, 128)},
    16384: {0.4: (2048, 8), 0.6: (1024, 16), 0.8: (512, 32), 0.85: (512, 32), 0.9: (256, 64), 0.95: (128, 128)},
    32768: {0.4: (4096, 8), 0.6: (2048, 16), 0.8: (1024, 32), 0.85: (1024, 32), 0.9: (512, 64), 0.95: (256, 128)},
    65536: {0.3: (8192, 8), 0.6: (4096, 16), 0.8: (2048, 32), 0.85: (1024, 64), 0.9: (1024, 64), 0.95: (512, 128)},
}
_TOLERANCE = 0.05


def lsh_threshold(b: int, r: int) -> float:
    """Similarity at which the S-curve is steepest: (1/b)^(1/r)."""
    return (1.0 / b) ** (1.0 / r)


def collision_probability(similarity: float, b: int, r: int) -> float:
    return 1.0 - (1.0 - similarity ** r) ** b


def false_rates(b: int, r: int, threshold: float) -> Tuple[float, float]:
    """Areas under the S-curve left of the threshold (false positives) and above it right of the
    threshold (false negatives), by adaptive quadrature as the reference does (br.py:162-220)."""
    from scipy.integrate import quad

    fp, _ = quad(lambda s: 1.0 - (1.0 - s ** r) ** b, 0.0, threshold, limit=100)
    fn, _ = quad(lambda s: (1.0 - s ** r) ** b, threshold, 1.0, limit=100)
    return fp, fn


def _search(num_perm: int, target: float) -> Optional[Tuple[int, int]]:
    best, best_score = None, math.inf
    root = int(math.sqrt(num_perm))
    # same visiting order as the reference (small r first, then small b): ties keep the first seen
    pairs = [(num_perm // r, r) for r in range(1, root + 1) if num_perm % r == 0]
    pairs += [(b, num_perm // b) for b in range(1, root + 1) if num_perm % b == 0]
    for b, r in pairs:
        if abs(lsh_threshold(b, r) - target) > _TOLERANCE:
            continue
        fp, fn = false_rates(b, r, target)
        if fp + fn < best_score:
            best, best_score = (b, r), fp + fn
    return best


def get_optimal_config(num_perm: int, target_threshold: float = 0.5) -> Tuple[int, int]:
    """(num_bands, rows_per_band) with num_bands * rows_per_band == num_perm."""
    table = _TABLE.get(num_perm)
    if table is not None:
        nearest = min(table, key=lambda t: abs(t - target_threshold))
        if abs(nearest - target_threshold) <= _TOLERANCE:
            return table[nearest]
    found = _search(num_perm, target_threshold)
    if found:
        return found
    b = int(math.sqrt(num_perm))
    while num_perm % b:
        b -= 1
    return b, num_perm // b
