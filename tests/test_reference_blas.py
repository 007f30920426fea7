"""`LSHHasher(reference_blas=...)`: keys pinned to a NAMED build of OpenBLAS.  No GPU: the named models
against NumPy running that very build (``OPENBLAS_CORETYPE`` in a subprocess), the constructor's contract, persistence."""

from __future__ import annotations

import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

from lshrs_amd import LSHRS, InMemoryStorage, LSHHasher, _hostblas

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# every row kind of sgemv_t (bands of 16, 5, 6, 7, 3 rows), the scalar tail (dim % 4 = 1, 2, 3), 8 m + 4 elements, blocks of 4096,
# and sdot (one row per band) at lengths with and without whole 32 / 64-element steps and a tail behind them
SHAPES = [(16, 768), (32, 1536), (4, 128), (5, 100), (6, 101), (7, 102), (3, 103), (13, 640), (10, 300), (7, 12), (4, 9), (2, 10),
          (2, 8), (5, 8),
          # fewer than 9 elements (round 5): the Haswell / Zen build runs its usual kernels there - modelled down to ONE element, every
          # row kind; the SkylakeX build takes small-matrix paths of its own - claimed by nobody
          (2, 2), (3, 3), (4, 5), (5, 6), (6, 7), (7, 1), (16, 4), (13, 3), (9, 7), (2, 5),
          (16, 2), (17, 3), (24, 6), (33, 7), (20, 8), (9, 5), (64, 7), (31, 4), (15, 8), (6, 3), (7, 6),
          (16, 4100 - 4), (4, 8192),
          # 8 m + 4 elements beyond 4096 (round 5): the short last block takes ITS first four first; with a scalar tail behind it
          (4, 4100), (5, 4108), (3, 8196), (6, 4101), (7, 8199), (2, 5004), (1, 768), (1, 64), (1, 100), (1, 96), (1, 33), (1, 31), (1, 7), (1, 1), (1, 1000), (1, 4100)]

_PROBE = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from lshrs_amd import _hostblas
build, shapes = sys.argv[1], json.loads(sys.argv[2])
lib = _hostblas.load()
rng = np.random.default_rng(77)
out = []
for r, dim in shapes:
    model = _hostblas.named_model(build, r, dim)
    planes = rng.standard_normal((3, r, dim)).astype(np.float32)
    licensed = int(_hostblas.blas_order_model(planes))
    bad = 0
    if model:
        for t in range(24):
            b = t %% 3
            x = rng.standard_normal(dim).astype(np.float32)
            if t %% 2 and dim > 1:                    # cancel against one row: only the order is left of y
                p = planes[b, t %% r].astype(np.float64)
                x = (x - (x @ p) / (p @ p) * p).astype(np.float32)
            want = planes[b] @ x                     # the reference's call, on THIS build (lshrs/hash/lsh.py:200)
            got = np.array([lib.lshrs_tb_model_row_dot(planes[b, i].ctypes.data, x.ctypes.data, dim, model, i, r)
                            for i in range(r)], dtype=np.float32)
            bad += int((want.view(np.uint32) != got.view(np.uint32)).sum())
    out.append([r, dim, model, licensed, bad])
print(json.dumps(out))
"""


@pytest.mark.parametrize("build,coretype", [("openblas-skylakex", "SkylakeX"), ("openblas-haswell", "Haswell"),
                                            ("openblas-haswell", "Zen"), ("openblas-skylakex", "Cooperlake")])
def test_named_model_is_numpy_on_that_build_bit_for_bit(build, coretype):
    """NumPy's bundled OpenBLAS forced onto the named build's kernels: wherever `named_model` claims a shape, the model
    reproduces `P_band @ x` bit for bit on every row of the band (random and cancelling vectors), and the licence check
    (`blas_order_model`) recognises an order too."""
    if _hostblas.numpy_blas() is None:
        pytest.skip("this NumPy is not built on OpenBLAS")
    env = dict(os.environ, OPENBLAS_CORETYPE=coretype, PYTHONPATH=ROOT)
    res = subprocess.run([sys.executable, "-c", _PROBE % {"root": ROOT}, build, json.dumps(SHAPES)], env=env,
                         capture_output=True, text=True, check=True)
    rows = json.loads(res.stdout.strip().splitlines()[-1])
    assert len(rows) == len(SHAPES)
    for r, dim, model, licensed, bad in rows:
        if dim < 9 and r > 1 and build == "openblas-skylakex":
            assert model == 3, (r, dim)                       # that build's small-matrix kernels (round 5: found and modelled)
        else:
            assert model in (1, 2), (r, dim)                  # every shape of the list is one the named builds are modelled for
        assert bad == 0, (build, r, dim, model, bad)
        assert licensed in (1, 2, 3), (build, r, dim)
    # the two builds are told apart where they differ: the scalar tail and sdot's kernel
    by = {(r, dim): model for r, dim, model, _, _ in rows}
    want = _hostblas.NAMED_BUILDS[build]
    assert by[(6, 101)] == by[(7, 102)] == by[(3, 103)] == by[(1, 768)] == by[(1, 100)] == want
    assert by[(16, 768)] == by[(10, 300)] == 1


def test_named_model_coverage_and_constructor_contract():
    nm = _hostblas.named_model
    assert nm("openblas-skylakex", 16, 768) == 1 and nm("openblas-haswell", 16, 768) == 1
    assert nm("openblas-skylakex", 16, 102) == 1 and nm("openblas-haswell", 16, 102) == 2 and nm("openblas-zen", 16, 102) == 0
    assert nm("openblas-skylakex", 1, 5) == 1 and nm("openblas-haswell", 1, 5) == 2
    assert nm("openblas-skylakex", 2, 8) == 3 and nm("openblas-skylakex", 4, 4) == 3 and nm("openblas-skylakex", 3, 5) == 3   # small-matrix kernels
    assert nm("openblas-haswell", 4, 4) == 1 and nm("openblas-haswell", 3, 5) == 2 and nm("openblas-haswell", 7, 1) == 2         # (round 5)
    assert nm("openblas-haswell", 2, 8) == 1 and nm("openblas-haswell", 16, 8) == 1             # (eight elements: the 8-lane kernels)
    assert nm("openblas-skylakex", 4, 4100) == 1 and nm("openblas-haswell", 6, 4101) == 2  # 8 m + 4 behind a full block (round 5)
    assert nm("mkl", 16, 768) == 0 and nm("host", 16, 768) == 0
    with pytest.raises(ValueError, match="reference_blas must be"):
        LSHHasher(16, 16, 768, reference_blas="mkl")
    with pytest.raises(ValueError, match="openblas-haswell.*Zen 4"):          # (round 6: the alias named the wrong kernels for Zen 4 / 5)
        LSHHasher(16, 16, 768, reference_blas="openblas-zen")
    assert LSHHasher(4, 4, 6, reference_blas="openblas-skylakex")._replay_model() == 3     # (model 3 covers dim <= 8, r >= 2)
    assert LSHHasher(4, 2, 8, reference_blas="openblas-skylakex")._replay_model() == 3      # (round 4: refused)
    with pytest.raises(ValueError, match="tie_replay"):
        LSHHasher(16, 16, 768, reference_blas="openblas-haswell", tie_replay="off")
    h = LSHHasher(16, 16, 768, seed=42, reference_blas="openblas-haswell")
    assert h.reference_blas == "openblas-haswell" and h._replay_model() == 1
    assert LSHHasher(16, 16, 102, reference_blas="openblas-haswell")._replay_model() == 2
    assert LSHHasher(64, 1, 100, reference_blas="openblas-skylakex")._replay_model() == 1


def test_a_named_build_keeps_the_device_route_on_a_host_whose_blas_is_not_recognised(monkeypatch):
    """What the 37x cliff of round 4 was: `blas_order_model` == 0 (MKL, BLIS, aarch64 ...) sends every batch to the host
    engine.  With a named build the route is the device's whatever the host answers."""
    monkeypatch.setattr(_hostblas, "blas_order_model", lambda planes: 0)
    ok = dict(aligned=True, short_stride=True, host_rows=False)
    assert LSHHasher(16, 16, 768, seed=42)._route(1_000_000, "host", **ok) == ("plain", 0)
    for build in ("openblas-skylakex", "openblas-haswell"):
        h = LSHHasher(16, 16, 768, seed=42, reference_blas=build)
        assert h._route(1_000_000, "host", **ok) == ("split+replay", 1)
        assert h._route(100, "host", **ok) == ("f32+replay", 1)
        assert not h._host_blas_agrees()                      # ... and the live audit against NumPy is off: NumPy is another BLAS
        assert LSHHasher(16, 16, 102, seed=42, reference_blas=build)._route(50_000, "host", **ok)[0] == "split+replay"   # (round 5: a scalar tail)
        assert LSHHasher(4, 13, 1001, seed=42, reference_blas=build)._route(50_000, "host", **ok)[0] == "f32+replay"
        assert LSHHasher(64, 1, 100, seed=42, reference_blas=build)._route(50_000, "host", **ok)[0] == "split+replay"    # (round 6: one-row bands too)
        assert LSHHasher(64, 1, 5, seed=42, reference_blas=build)._route(50_000, "host", **ok)[0] == "f32+replay"


def test_the_choice_travels_with_the_index(tmp_path):
    idx = LSHRS(dim=64, num_perm=32, storage=InMemoryStorage(), reference_blas="openblas-haswell")
    assert idx._hasher.reference_blas == "openblas-haswell"
    idx.save_to_disk(tmp_path / "idx")
    meta = json.load(open(tmp_path / "idx" / "metadata.json"))
    assert meta["lshrs_amd"] == {"reference_blas": "openblas-haswell"} and set(meta) == {"version", "config", "redis_config", "lshrs_amd"}
    meta["lshrs_amd"]["reference_blas"] = "openblas-zen"          # an index saved by round 5 under the alias: read as what it computed
    json.dump(meta, open(tmp_path / "idx" / "metadata.json", "w"))
    assert LSHRS.load_from_disk(tmp_path / "idx", storage=InMemoryStorage())._hasher.reference_blas == "openblas-haswell"
    back = LSHRS.load_from_disk(tmp_path / "idx", storage=InMemoryStorage())
    assert back._hasher.reference_blas == "openblas-haswell"
    assert all(np.array_equal(a, b) for a, b in zip(back._hasher.projections, idx._hasher.projections))
    again = pickle.loads(pickle.dumps(idx))
    assert again._hasher.reference_blas == "openblas-haswell"
    assert pickle.loads(pickle.dumps(idx._hasher)).reference_blas == "openblas-haswell"
    # the default records which build the host's NumPy ran (round 6) under a key of our own; the reference's three keys are as before
    plain = LSHRS(dim=64, num_perm=32, storage=InMemoryStorage())
    plain.save_to_disk(tmp_path / "plain")
    pmeta = json.load(open(tmp_path / "plain" / "metadata.json"))
    assert set(pmeta) == {"version", "config", "redis_config", "lshrs_amd"}
    assert pmeta["lshrs_amd"] == {"reference_blas": "host", "host_blas": LSHHasher.host_blas_name()}
    assert LSHRS.load_from_disk(tmp_path / "plain", storage=InMemoryStorage())._hasher.reference_blas == "host"
    old = LSHHasher(4, 4, 32).__getstate__()
    old.pop("reference_blas")
    old.pop("_host_agrees")
    h = LSHHasher.__new__(LSHHasher)
    h.__setstate__(old)
    assert h.reference_blas == "host"


def test_sdot_roundings_bound_the_hosts_value_for_mass_in_the_last_elements():
    """the proven window of a ONE-row band must charge sdot's roundings, not sgemv's - a vector whose mass sits in the
    last elements passes more roundings than the sgemv count gives.  `host_roundings_sdot` as an elementwise bound:
    |y_model - y_exact| <= 2^-24 sum_k m[k] |p_k x_k| for both builds, random rows and rows with the mass at either end."""
    from lshrs_amd.windows import host_roundings, host_roundings_sdot, window_coefficients

    lib = _hostblas.load()
    rng = np.random.default_rng(8)
    u = 2.0 ** -24
    for dim in (64, 96, 100, 768, 777, 31, 8, 1):
        K = (dim + 31) // 32 * 32
        m = host_roundings_sdot(dim, K)[:dim]
        assert m.min() >= 3 and (dim < 64 or m[dim - 1] >= 3)
        for t in range(60):
            p = rng.standard_normal(dim).astype(np.float32)
            x = rng.standard_normal(dim).astype(np.float32)
            if t % 3 == 1:
                x[:max(0, dim - 8)] *= np.float32(1e-6)       # the mass on the last 8 elements
            elif t % 3 == 2:
                x[8:] *= np.float32(1e-6)                     # ... on the first 8
            exact = float(p.astype(np.float64) @ x.astype(np.float64))
            budget = u * float(m @ np.abs(p.astype(np.float64) * x.astype(np.float64))) * (1 + 1e-6)
            for model in (1, 2):
                y = float(lib.lshrs_tb_model_row_dot(p.ctypes.data, x.ctypes.data, dim, model, 0, 1))
                assert abs(y - exact) <= budget, (dim, t, model, abs(y - exact) / budget)
    # at the last elements of a long row the sdot count is what the window now uses - and it is above the sgemv single-row count there
    dim, K = 768, 768
    sd, sg = host_roundings_sdot(dim, K), host_roundings(dim, K, np.array([2]))[0]
    assert sd[-1] == 10 and sg[-1] == 5 and (sd >= 3).all()      # (SkylakeX build: own step 1 + fold, turn-adds, lanes, pairs, final = 9)
    planes = rng.standard_normal((8, dim)).astype(np.float32)
    ca1, _, ct1, _ = window_coefficients(planes, 1, 1)
    assert np.isfinite(ca1).all() and (ca1 > 0).all() and (ct1 > 0).all()


_SAVE = r"""
import sys
sys.path.insert(0, %(root)r)
from lshrs_amd import LSHRS, InMemoryStorage, LSHHasher
import pickle
print("BUILD", LSHHasher.host_blas_name())
for dim in (102, 768):
    idx = LSHRS(dim=dim, num_perm=256, num_bands=16, rows_per_band=16, storage=InMemoryStorage())
    idx.save_to_disk(sys.argv[1] + "/idx%%d" %% dim)
    pickle.dump(idx, open(sys.argv[1] + "/idx%%d.pkl" %% dim, "wb"))
"""

_LOAD = r"""
import json, pickle, sys, warnings
sys.path.insert(0, %(root)r)
from lshrs_amd import LSHRS, InMemoryStorage, LSHHasher
out = {"build": LSHHasher.host_blas_name()}
for dim in (102, 768):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        idx = LSHRS.load_from_disk(sys.argv[1] + "/idx%%d" %% dim, storage=InMemoryStorage())
        out["disk%%d" %% dim] = [str(x.message) for x in w if x.category.__name__ == "ReferenceBlasMismatch"]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        pickle.load(open(sys.argv[1] + "/idx%%d.pkl" %% dim, "rb"))
        out["pickle%%d" %% dim] = [str(x.message) for x in w if x.category.__name__ == "ReferenceBlasMismatch"]
try:
    LSHRS.load_from_disk(sys.argv[1] + "/idx102", storage=InMemoryStorage(), strict_blas=True)
    out["strict"] = "loaded"
except ValueError as exc:
    out["strict"] = str(exc)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    pinned = LSHRS.load_from_disk(sys.argv[1] + "/idx102", storage=InMemoryStorage(), reference_blas="openblas-skylakex")
    out["pinned"] = [pinned._hasher.reference_blas, len(w)]
print(json.dumps(out))
"""


def test_host_blas_name_and_the_warning_when_a_host_index_moves_to_the_other_build(tmp_path):
    """`reference_blas="host"` on an AVX-512 host (OpenBLAS's SkylakeX-class kernels - Zen 4 / 5 included) and on
    an AVX2 host are different keys at shapes where the two builds sum differently.  `save_to_disk` / pickle record which build the
    host ran; loading on the other one warns (naming both) at 16 x 16 x 102 (a scalar tail) and is silent at 16 x 16 x 768."""
    if _hostblas.numpy_blas() is None:
        pytest.skip("this NumPy is not built on OpenBLAS")
    run = lambda script, core: subprocess.run([sys.executable, "-c", script % {"root": ROOT}, str(tmp_path)],   # noqa: E731
                                               env=dict(os.environ, OPENBLAS_CORETYPE=core, PYTHONPATH=ROOT),
                                               capture_output=True, text=True, check=True).stdout
    assert "BUILD openblas-skylakex" in run(_SAVE, "SkylakeX")
    assert json.load(open(tmp_path / "idx102" / "metadata.json"))["lshrs_amd"] == {"reference_blas": "host", "host_blas": "openblas-skylakex"}
    same = json.loads(run(_LOAD, "SkylakeX").strip().splitlines()[-1])
    assert same["build"] == "openblas-skylakex" and same["disk102"] == same["pickle102"] == same["disk768"] == [] and same["strict"] == "loaded"
    other = json.loads(run(_LOAD, "Haswell").strip().splitlines()[-1])
    assert other["build"] == "openblas-haswell"
    for key in ("disk102", "pickle102"):
        assert len(other[key]) == 1 and "openblas-skylakex" in other[key][0] and "openblas-haswell" in other[key][0], other
    assert other["disk768"] == other["pickle768"] == []                        # (whole groups of four, bands of 16 rows: both builds sum alike)
    assert "openblas-skylakex" in other["strict"] and "openblas-haswell" in other["strict"]
    assert other["pinned"] == ["openblas-skylakex", 0]                         # (naming the recorded build: hashed as built, nothing to warn about)
    assert _hostblas.builds_differ(16, 102) and _hostblas.builds_differ(1, 768) and _hostblas.builds_differ(4, 6)
    assert not _hostblas.builds_differ(16, 768) and not _hostblas.builds_differ(32, 1536) and not _hostblas.builds_differ(4, 128)


def test_an_unrecognised_host_blas_says_so_once(monkeypatch):
    """a NumPy / OpenBLAS whose summation order the replay does not know drops the hasher from the device
    route to the host engine (same keys, ~40 x slower) - it says so, once per hasher (`HostBlasNotRecognised`); a named build,
    and `tie_replay="off"`, have nothing to say."""
    import warnings

    from lshrs_amd import HostBlasNotRecognised, LSHHasher, _hostblas

    monkeypatch.setattr(_hostblas, "blas_order_model", lambda planes: 0)
    h = LSHHasher(8, 8, 64, seed=1)
    with pytest.warns(HostBlasNotRecognised, match="not one the device replay knows"):
        assert h._replay_model() == 0
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert h._replay_model() == 0                      # (cached; and once per hasher anyway)
        h._replay_model_cache = None
        assert h._replay_model() == 0
        assert LSHHasher(8, 8, 64, seed=1, reference_blas="openblas-haswell")._replay_model() != 0
        assert LSHHasher(8, 8, 64, seed=1, tie_replay="off")._replay_model() == 0
