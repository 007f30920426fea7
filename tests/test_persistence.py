"""On-disk / pickle persistence of ``LSHRS``: same format as the reference
(tests/golden/g7_saved_index was written by the reference's own ``save_to_disk``).  CPU only — building
a hasher and assigning hyperplanes touches no device."""

from __future__ import annotations

import json
import os
import pickle

import numpy as np
import pytest

from lshrs_amd import LSHRS, InMemoryStorage, LSHHasher
from oracle.lshrs_oracle import make_projections


def test_loads_an_index_saved_by_the_reference(golden_dir):
    src = os.path.join(golden_dir, "g7_saved_index")
    idx = LSHRS.load_from_disk(src, storage=InMemoryStorage(), redis_config={"password": "pw"})
    assert type(idx._hasher) is LSHHasher
    assert idx._config == {"dim": 32, "num_perm": 16, "num_bands": 4, "rows_per_band": 4,
                           "similarity_threshold": 0.6, "buffer_size": 77, "seed": 9}
    assert idx._redis_config["prefix"] == "pfx" and idx._redis_config["password"] == "pw"
    want = make_projections(4, 4, 32, 9)            # what the reference drew for seed 9 and then saved
    with np.load(os.path.join(src, "projections.npz")) as data:
        assert sorted(data.files) == [f"arr_{i}" for i in range(4)]
        for i in range(4):
            assert np.array_equal(data[f"arr_{i}"], want[i])
            assert np.array_equal(idx._hasher.projections[i], want[i]) and idx._hasher.projections[i].dtype == np.float32
    with pytest.raises(FileNotFoundError):
        LSHRS.load_from_disk(os.path.join(golden_dir, "nope"), storage=InMemoryStorage())


def test_save_writes_the_reference_format(tmp_path, golden_dir):
    idx = LSHRS(dim=32, num_bands=4, rows_per_band=4, num_perm=16, buffer_size=77, seed=9, storage=InMemoryStorage(),
                redis_password="hunter2", redis_prefix="pfx", similarity_threshold=0.6)
    out = tmp_path / "saved"
    idx.save_to_disk(out)
    assert sorted(os.listdir(out)) == ["metadata.json", "projections.npz"]
    mine = json.load(open(out / "metadata.json"))
    ref = json.load(open(os.path.join(golden_dir, "g7_saved_index", "metadata.json")))
    # (round 6: one key of our own beside the reference's three - which BLAS build the host's NumPy ran; the reference's loader
    #  reads "config" / "redis_config" only, lshrs/core/main.py:941-947)
    assert mine.pop("lshrs_amd") == {"reference_blas": "host", "host_blas": LSHHasher.host_blas_name()}
    assert mine == ref                                  # incl. version string and "<REDACTED>" password
    assert "hunter2" not in open(out / "metadata.json").read()
    with np.load(out / "projections.npz") as a, np.load(os.path.join(golden_dir, "g7_saved_index", "projections.npz")) as b:
        assert a.files == b.files
        assert all(np.array_equal(a[k], b[k]) for k in a.files)
    back = LSHRS.load_from_disk(out, storage=InMemoryStorage())
    assert back._config == idx._config
    assert all(np.array_equal(p, q) for p, q in zip(back._hasher.projections, idx._hasher.projections))


def test_reassigned_hyperplanes_survive_the_round_trip(tmp_path):
    idx = LSHRS(dim=16, num_bands=2, rows_per_band=8, num_perm=16, storage=InMemoryStorage(), seed=1)
    custom = [np.full((8, 16), i + 0.5, dtype=np.float32) for i in range(2)]
    v0 = idx._hasher._projection_version
    idx._hasher.projections = custom
    assert idx._hasher._projection_version > v0         # device image would be rebuilt on next use
    idx.save_to_disk(tmp_path / "x")
    back = LSHRS.load_from_disk(tmp_path / "x", storage=InMemoryStorage())
    assert all(np.array_equal(a, b) for a, b in zip(back._hasher.projections, custom))


def test_pickle_round_trip_flushes_and_keeps_hyperplanes(monkeypatch):
    import lshrs_amd.core as core

    monkeypatch.setattr(core, "default_storage", lambda **kw: InMemoryStorage(prefix=kw["prefix"]))
    from tests._doubles import OracleBackedHasher

    store = InMemoryStorage()
    idx = LSHRS(dim=32, num_bands=4, rows_per_band=4, num_perm=16, storage=store, seed=3, buffer_size=1000,
                hasher=OracleBackedHasher(4, 4, 32, 3), vector_fetch_fn=lambda ids: None, redis_prefix="zz")
    idx.ingest(5, np.random.default_rng(0).standard_normal(32).astype(np.float32))
    assert len(idx._buffer) == 4
    blob = pickle.dumps(idx)
    assert idx._buffer == [] and store.total_operations == 4          # __getstate__ flushes (main.py:1003)
    clone = pickle.loads(blob)
    assert clone._config == idx._config and clone._redis_config["prefix"] == "zz"
    assert clone._vector_fetch_fn is None and clone._buffer == []
    assert all(np.array_equal(a, b) for a, b in zip(clone._hasher.projections, idx._hasher.projections))


def test_delete_and_clear_follow_the_reference():
    from tests._doubles import OracleBackedHasher

    store = InMemoryStorage()
    idx = LSHRS(dim=32, num_bands=4, rows_per_band=4, num_perm=16, storage=store, buffer_size=1000,
                hasher=OracleBackedHasher(4, 4, 32, 42))
    data = np.random.default_rng(1).standard_normal((3, 32)).astype(np.float32)
    for i in range(3):
        idx.ingest(i, data[i])
    idx.delete(1)                                       # does not flush (main.py:783-784)
    assert len(idx._buffer) == 12 and store.total_operations == 0
    idx.clear()                                         # flushes first, then clears the buckets (main.py:795-796)
    assert idx._buffer == [] and store.total_operations == 12 and store._buckets == {}
    assert idx.stats() == {"dimension": 32, "num_perm": 16, "num_bands": 4, "rows_per_band": 4, "buffer_size": 1000,
                           "similarity_threshold": 0.5, "redis_prefix": "lsh"}


def test_pickle_carries_device_windows_and_ingest_mode_and_defers_the_storage():
    """What this build adds to the pickle (under a key the reference's __setstate__ never reads): the GPU index, the
    window modes, the ingest mode.  Unpickling must not need the Redis client: the storage is resolved on first use."""
    from lshrs_amd import LSHHasher

    idx = LSHRS(dim=64, num_bands=8, rows_per_band=16, num_perm=128, storage=InMemoryStorage(), seed=5,
                packed_ingest=True, hasher=LSHHasher(8, 16, 64, seed=5, device=3, tau1_ulps="bound", tau_ulps=16.0,
                                                     margin_guard=0.25, tie_replay="off"))
    state = idx.__getstate__()
    assert set(state) >= {"config", "redis_config", "projections"}            # the reference's three keys, untouched
    clone = pickle.loads(pickle.dumps(idx))                                    # (no redis in this image: must not raise)
    h = clone._hasher
    assert clone._packed_ingest is True and h._device == 3 and h.window_mode == {"tau": "measured", "tau1": "bound"}
    assert h.tau_ulps == 16.0 and h.tau1_ulps == idx._hasher.tau1_ulps and h.margin_guard == 0.25 and h.tie_replay == "off"
    assert all(np.array_equal(a, b) for a, b in zip(h.projections, idx._hasher.projections))
    assert not hasattr(clone._storage, "batch_add_csr")                         # capability probes do not open a connection
    with pytest.raises(RuntimeError, match="RedisStorage"):
        clone._storage.get_bucket(0, b"\x00\x00")                              # first real use resolves it (and says why it cannot)
    # a state written by the reference (no extra key) still loads
    del state["lshrs_amd"]
    plain = LSHRS.__new__(LSHRS)
    plain.__setstate__(state)
    assert plain._packed_ingest == "auto" and plain._hasher.window_mode["tau1"] == "bound"       # (the default: the proven window)
    # a storage proxy whose __init__ has not run (copy / pickle build instances that way) must not recurse
    import copy

    from lshrs_amd.core import _DeferredStorage

    bare = _DeferredStorage.__new__(_DeferredStorage)
    with pytest.raises(AttributeError):
        bare._real
    assert copy.copy(clone._storage)._cfg == clone._storage._cfg
