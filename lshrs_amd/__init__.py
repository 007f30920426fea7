"""lshrs_amd — MI355X-native implementation of the lshrs compute hot path.

Drop-in names (same surface as the reference package ``lshrs``):

    LSHRS / lshrs, LSHHasher, HashSignatures, top_k_cosine, cosine_similarity, l2_norm

The arithmetic lives in ``csrc/*.hip`` (gfx950 only; one translation unit per kernel family) behind the C ABI of
``include/lshrs_hip.h``; this package is the Python boundary around it.  There is
no CPU compute path: without the built extension and a visible MI355X every
compute call raises ``lshrs_amd._native.NativeLibraryError``.
"""

from ._config import HashSignatures
from ._native import NativeLibraryError
from .bandrows import get_optimal_config
from .core import LSHRS, lshrs
from .hasher import HostBlasNotRecognised, LSHHasher
from .packed_ops import RedisPackedWriter, group_by_bucket, hex_keys
from .similarity import cosine_similarity, l2_norm, rerank_batch, top_k_cosine
from .storage import BucketOperation, InMemoryStorage

__all__ = [
    "LSHRS", "lshrs", "LSHHasher", "HashSignatures", "top_k_cosine", "cosine_similarity", "l2_norm",
    "rerank_batch", "get_optimal_config", "InMemoryStorage", "BucketOperation", "NativeLibraryError",
    "RedisPackedWriter", "group_by_bucket", "hex_keys", "HostBlasNotRecognised",
]
__version__ = "0.1.0"
