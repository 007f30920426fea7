"""Shared pytest configuration.

Markers
  gpu   needs a real MI355X (run with ``-m gpu`` on the GPU box; skipped where /dev/kfd is absent)
  perf  wall-clock floors on a real MI355X (``-m perf``): kept OUT of ``-m gpu`` so that a slow or shared box cannot turn a
        parity run red for a non-parity reason; skipped where /dev/kfd is absent
"""

from __future__ import annotations

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "perf: wall-clock floor on a real MI355X (run with -m perf; not part of -m gpu)")


def pytest_collection_modifyitems(config, items):
    if os.path.exists("/dev/kfd"):
        return
    skip = pytest.mark.skip(reason="no AMD GPU device node (/dev/kfd) in this container")
    for item in items:
        if "gpu" in item.keywords or "perf" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def rng() -> np.random.Generator:
    return np.random.default_rng(12345)


@pytest.fixture(scope="session")
def golden_dir() -> str:
    return GOLDEN
