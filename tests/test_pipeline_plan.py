"""Host logic of the chunked signature path (no GPU): the plan `LSHHasher._pipeline_plan` hands to either driver."""

from __future__ import annotations

import numpy as np
import pytest

from lshrs_amd import LSHHasher


def _plan(n, chunk=262_144, pair_head=True):
    h = LSHHasher(16, 16, 768, seed=1)
    h.pipeline_chunk_rows = chunk
    h.pipeline_pair_head = pair_head
    return h._pipeline_plan(n)


@pytest.mark.parametrize("n", [131_072, 131_073, 200_000, 300_000, 524_288, 856_432, 1_000_000, 1_048_576, 1_250_000,
                               5_000_000, 10_000_019])
def test_plan_covers_the_batch_in_order(n):
    ch, cap, spans = _plan(n)
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))           # contiguous, ascending
    sizes = [hi - lo for lo, hi in spans]
    assert all(0 < s <= ch for s in sizes)
    assert cap == ch // 32 + 1024                                          # room for ~12x the measured tie rate
    # nothing overlaps the host work of the last chunk: it is at most one full-chip round, unless the batch is
    # too short to cut one off
    assert sizes[-1] <= 65_536 or len(spans) <= 4 and sizes[-1] < 131_072


def test_bench_batch_is_four_chunks_with_a_paired_head():
    ch, cap, spans = _plan(1_000_000)
    assert [hi - lo for lo, hi in spans] == [524_288, 262_144, 148_032, 65_536]
    assert (ch, cap) == (524_288, 17_408)
    # the knob restores one launch per chunk
    assert [hi - lo for lo, hi in _plan(1_000_000, pair_head=False)[2]] == [262_144] * 3 + [148_032, 65_536]


def test_long_batches_keep_one_single_chunk_before_the_short_tail():
    sizes = [hi - lo for lo, hi in _plan(5_000_000)[2]]
    assert sizes == [524_288] * 9 + [262_144, 19_264]
    sizes = [hi - lo for lo, hi in _plan(1_250_000)[2]]
    assert sizes == [524_288, 262_144, 262_144, 135_888, 65_536]
    # the last full-size chunk is never part of a pair: its ties are resolved while the short chunks run
    for n in (1_000_000, 1_250_000, 3_000_000, 10_000_019):
        sizes = [hi - lo for lo, hi in _plan(n)[2]]
        full = [i for i, s in enumerate(sizes) if s in (262_144, 524_288)]
        assert sizes[full[-1]] == 262_144


def test_mid_size_batches_take_smaller_chunks():
    assert [hi - lo for lo, hi in _plan(300_000)[2]] == [131_072, 131_072, 37_856]
    assert [hi - lo for lo, hi in _plan(131_073)[2]] == [65_536, 65_536, 1]
    assert _plan(300_000, chunk=131_072)[1] == 131_072 // 32 + 1024


def test_plan_does_not_depend_on_data_or_device():
    a = _plan(2_000_000)
    b = _plan(2_000_000)
    assert a == b and isinstance(a[2], list) and all(isinstance(v, int) for s in a[2] for v in s)
    assert np.sum([hi - lo for lo, hi in a[2]]) == 2_000_000
