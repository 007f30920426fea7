"""Value type at the hasher boundary.

Mirrors the observable behaviour of the reference's ``HashSignatures``
(lshrs/_config/config.py:12-71): a frozen, hashable, iterable, indexable wrapper
around ``tuple[bytes, ...]`` with one packed key per band, in band order.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Iterator, Tuple


@dataclass(frozen=True)
class HashSignatures:
    """Per-vector LSH band keys (one ``bytes`` per band)."""

    bands: Tuple[bytes, ...]

    def __post_init__(self) -> None:
        # Whatever sequence of bytes-likes came in, hold an immutable tuple of bytes
        # (reference: config.py:36-41).
        object.__setattr__(self, "bands", tuple(bytes(b) for b in self.bands))

    def __iter__(self) -> Iterator[bytes]:
        return iter(self.bands)

    def __len__(self) -> int:
        return len(self.bands)

    def __getitem__(self, item: int) -> bytes:
        return self.bands[item]

    def as_tuple(self) -> Tuple[bytes, ...]:
        return self.bands
