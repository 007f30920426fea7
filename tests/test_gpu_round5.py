"""GPU tests of round 5.  Every call goes through the C ABI; the checker is the oracle's literal restatement of
lshrs/hash/lsh.py:96-211."""

from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def _hasher(seed, nb, r, dim, **kw):
    from lshrs_amd import LSHHasher

    return LSHHasher(num_bands=nb, rows_per_band=r, dim=dim, seed=seed, **kw)


# ----------------------------------------------------------------------------- one pass as a pipeline
@pytest.mark.parametrize("nb,r,dim,n,seed", [(16, 16, 768, 450_000, 42), (16, 32, 1536, 300_000, 7), (20, 10, 768, 420_000, 3)])
def test_chunked_pass_gives_the_keys_of_the_one_launch_pass(torch_mod, nb, r, dim, n, seed):
    """ABI 6, `lshrs_sig_hash_batch_split_replay_chunked_f32`: stage 2 of a chunk beside the next chunk's stage 1.  Chunks are
    row ranges: the keys, the flagged projections and the sign flips are those of the one-launch pass; the audit still runs."""
    from oracle import lshrs_oracle as O

    torch = torch_mod
    h = _hasher(seed, nb, r, dim)
    if not h._replay_model():
        pytest.skip("the host BLAS's summation order is not one the replay knows on this box")
    x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(11))
    flags0 = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ref = h.hash_device(x, row_flags=flags0).clone()
    one = dict(h.last_stats)
    assert one["route"] == "split+replay" and "chunks" not in one
    rnd = 65_536 if nb * h.band_bytes * 8 <= 256 or nb * r <= 256 else 32_768
    for plan in ("on", [rnd, 2 * rnd, n - 3 * rnd], [n - 1000, 1000], [rnd] * 6 + [n - 6 * rnd]):
        h.chunking = plan
        flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
        keys = h.hash_device(x, row_flags=flags)
        st = dict(h.last_stats)
        assert st.get("chunks", 1) == (3 if plan == "on" else len(plan)), (plan, st)
        assert torch.equal(keys, ref) and torch.equal(flags, flags0), plan
        # (a chunk that starts inside a 256-row tile moves its rows to other lanes: the f32 sums of a row's two norms may differ
        #  in the last place, and a projection exactly at the window's edge be listed or not - either is inside the proof)
        assert st["relaunches"] == 0 and abs(st["flagged"] - one["flagged"]) <= 2 and st["sign_flips"] == one["sign_flips"], (st, one)
        assert sum(st["chunk_flagged"]) == st["flagged"]
        assert abs(st["tie_pairs"] - one["tie_pairs"]) <= 2
        assert st["audited_unflagged"] >= 0.7 * one["audited_unflagged"] and st["audit_sign_disagreements"] == 0, st
        assert 0.0 < st["max_dev_units"] <= h.window_info["window_units_worst_case_row"]
    rows = np.r_[0:700, rnd - 300:rnd + 300, n - 700:n]
    assert np.array_equal(ref[rows].cpu().numpy(), O.hash_batch_literal_packed(h.projections, x[rows].cpu().numpy()))
    # the streaming form takes the same path
    h.chunking = "on"
    assert torch.equal(h.hash_device_async(x).result(), ref)
    h.chunking = "off"


def test_chunked_pass_repeats_when_one_chunk_outgrows_its_share_of_the_list(torch_mod):
    """Rows flagged wholesale (largest |x| outside [2^-32, 2^32]) packed into ONE chunk: that chunk's share of the stage-1 list
    overflows although the batch's total would fit - the pass is repeated with room, and the keys are the reference's."""
    from oracle import lshrs_oracle as O

    torch = torch_mod
    h = _hasher(42, 16, 16, 768)
    if not h._replay_model():
        pytest.skip("the host BLAS's summation order is not one the replay knows on this box")
    n = 400_000
    x = torch.randn(n, 768, device="cuda", generator=torch.Generator("cuda").manual_seed(12))
    x[200_000:203_000] *= 2.0 ** 40            # 3 000 rows x 256 projections = 768 000 entries in the second chunk
    h.chunking = [131_072, 131_072, n - 262_144]
    keys = h.hash_device(x)
    st = dict(h.last_stats)
    assert st["relaunches"] >= 1 and st["chunks"] == 3 and st["flagged"] >= 768_000, st
    h.chunking = "off"
    assert torch.equal(keys, h.hash_device(x))
    rows = np.r_[199_500:203_500]
    assert np.array_equal(keys[rows].cpu().numpy(), O.hash_batch_literal_packed(h.projections, x[rows].cpu().numpy()))


def test_chunked_pass_guards_still_fire(torch_mod):
    """The audit of un-flagged projections sees every chunk: the adversarial rows of tests/_adversary.py behind a 64-unit
    measured window - placed in the LAST chunk - trip it in a chunked pass as in a one-launch pass (then: proven window, batch
    repeated, the reference's keys)."""
    from oracle.lshrs_oracle import hash_batch_literal_packed
    from tests._adversary import adversarial_row

    torch = torch_mod
    nb, r, dim, n, k = 16, 16, 768, 400_000, 2_048
    base = _hasher(42, nb, r, dim)
    if not base._replay_model():
        pytest.skip("the host BLAS's summation order is not one the replay knows on this box")
    adv = np.empty((k, dim), dtype=np.float32)
    for i in range(k):
        adv[i] = adversarial_row((1.0 if i % 2 else -1.0) * base.projections[(i // r) % nb][i % r], 20.0, seed=i)
    x = torch.randn(n, dim, device="cuda", generator=torch.Generator("cuda").manual_seed(13))
    x[300_000:300_000 + k] = torch.from_numpy(adv).cuda()
    h = _hasher(42, nb, r, dim, tau1_ulps=64.0, audit_unflagged=1_000_000)
    h.chunking = [131_072, 131_072, n - 262_144]
    launches = 0
    while h.window_mode["tau1"] == "measured" and launches < 200:
        out = h.hash_device(x)
        launches += 1
    st = dict(h.last_stats)
    assert h.window_mode["tau1"] == "bound" and st.get("audit_escalations", 0) >= 1, (launches, st)
    want = hash_batch_literal_packed(base.projections, adv)
    assert np.array_equal(out[300_000:300_000 + k].cpu().numpy(), want)
    assert torch.equal(out, base.hash_device(x))
