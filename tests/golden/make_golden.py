#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REAL REFERENCE.

Run only in the build container (needs /root/reference; the GPU box never has it):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What it does: imports the upstream lshrs modules *unmodified* from
/root/reference (the hot-path modules by file path, SURVEY.md §8c recipe 1; the
orchestrator through an in-memory stand-in for the absent ``redis`` client
library, recipe 2, written to a temp dir and never committed), feeds them seeded
inputs, and stores inputs' seeds + the reference's outputs as data:

  g1_projections.json   hyperplane digests per (seed, bands, rows, dim)
  g2_signatures.npz     packed band keys of 256 seeded vectors per config + f64 sign margins
  g3_specials.json      ±0 / NaN / ±Inf / subnormal inputs -> keys
  g4_cosine.npz/.json   cosine_similarity / top_k_cosine outputs
  g5_orchestration.json LSHRS.index batches, get_top_k / get_above_p results, error timing
  g6_autoconfig.json    get_optimal_config table
  g7_saved_index/       an index directory written by the reference's save_to_disk
  g8_queries_768.json   100 queries against 3 000 clustered 768-d vectors (num_perm 256): query() results

Fixtures are data only (inputs are regenerated from seeds by the tests).
"""

from __future__ import annotations

import hashlib
import importlib.util
import json
import os
import sys
import tempfile
import types

import numpy as np

REF_ROOT = "/root/reference"
REF = REF_ROOT + "/lshrs"
OUT = os.path.dirname(os.path.abspath(__file__))

sys.dont_write_bytecode = True


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m


def _load(name, file):
    spec = importlib.util.spec_from_file_location(name, file)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_hot_path():
    _pkg("lshrs", REF)
    _pkg("lshrs._config", REF + "/_config")
    _pkg("lshrs.hash", REF + "/hash")
    _pkg("lshrs.utils", REF + "/utils")
    cfg = _load("lshrs._config.config", REF + "/_config/config.py")
    lsh = _load("lshrs.hash.lsh", REF + "/hash/lsh.py")
    norm = _load("lshrs.utils.norm", REF + "/utils/norm.py")
    sim = _load("lshrs.utils.similarity", REF + "/utils/similarity.py")
    br = _load("lshrs.utils.br", REF + "/utils/br.py")
    return cfg, lsh, norm, sim, br


def pack_sigs(sigs, nb, bb):
    out = np.empty((len(sigs), nb, bb), dtype=np.uint8)
    for i, s in enumerate(sigs):
        for b, key in enumerate(s):
            out[i, b] = np.frombuffer(key, dtype=np.uint8)
    return out


SIG_CONFIGS = [  # (seed, bands, rows, dim, data_seed)
    (42, 16, 4, 128, 101),
    (42, 16, 16, 768, 102),
    (7, 16, 32, 1536, 103),
    (123, 3, 5, 4, 104),
    (42, 4, 12, 32, 105),
    (5, 2, 24, 100, 106),   # dim not a multiple of 32, rows_per_band -> 3 bytes
    (9, 5, 8, 30, 107),     # dim not a multiple of 4
]


def main() -> None:
    cfg, lsh, norm, sim, br = load_hot_path()

    # ---- G1 -----------------------------------------------------------------
    g1 = []
    for seed, nb, r, dim, _ in SIG_CONFIGS:
        h = lsh.LSHHasher(num_bands=nb, rows_per_band=r, dim=dim, seed=seed)
        stacked = np.concatenate(h.projections, axis=0)
        assert stacked.dtype == np.float32
        g1.append({
            "seed": seed, "num_bands": nb, "rows_per_band": r, "dim": dim,
            "sha256": hashlib.sha256(stacked.tobytes()).hexdigest(),
            "first8": [float(v) for v in stacked.reshape(-1)[:8]],
            "first8_hex": stacked.reshape(-1)[:8].tobytes().hex(),
        })
    json.dump(g1, open(os.path.join(OUT, "g1_projections.json"), "w"), indent=1)

    # ---- G2 -----------------------------------------------------------------
    g2 = {}
    for seed, nb, r, dim, dseed in SIG_CONFIGS:
        h = lsh.LSHHasher(num_bands=nb, rows_per_band=r, dim=dim, seed=seed)
        x = np.random.default_rng(dseed).standard_normal((256, dim)).astype(np.float32)
        sigs = h.hash_batch(x)
        packed = pack_sigs(sigs, nb, (r + 7) // 8)
        # single-vector path must agree with the batch path (it is the same code)
        assert all(h.hash_vector(x[i]).as_tuple() == sigs[i].as_tuple() for i in range(0, 256, 37))
        p64 = np.concatenate(h.projections, axis=0).astype(np.float64)
        y64 = x.astype(np.float64) @ p64.T
        tag = f"s{seed}_b{nb}_r{r}_d{dim}_x{dseed}"
        g2[tag + "_keys"] = packed
        g2[tag + "_minabs"] = np.abs(y64).min(axis=1)
    np.savez_compressed(os.path.join(OUT, "g2_signatures.npz"), **g2)

    # ---- G3 specials --------------------------------------------------------
    g3 = []
    h = lsh.LSHHasher(num_bands=2, rows_per_band=4, dim=4, seed=1)
    custom = [
        np.array([[1, -1, 0, 0], [-1, 1, 0, 0], [1, 1, 0, 0], [-1, -1, 0, 0]], dtype=np.float32),
        np.array([[0, 0, 1, 0], [0, 0, -1, 0], [0, 0, 0, 1], [1, 0, 0, -1]], dtype=np.float32),
    ]
    h.projections = custom
    sub = np.float32(1e-45)
    cases = {
        "plus_minus_zero": [1.0, 1.0, 0.0, 0.0],
        "all_zero": [0.0, 0.0, 0.0, 0.0],
        "neg_zero": [-0.0, -0.0, -0.0, -0.0],
        "nan_first": [float("nan"), 1.0, 2.0, 3.0],
        "nan_last": [1.0, 2.0, 3.0, float("nan")],
        "pos_inf": [float("inf"), 1.0, 1.0, 1.0],
        "neg_inf": [float("-inf"), 1.0, 1.0, 1.0],
        "inf_minus_inf": [float("inf"), float("inf"), 1.0, -1.0],
        "subnormal": [float(sub), 0.0, float(sub), -float(sub)],
        "tiny": [1e-9, -1e-9, 1e-9, 1e-9],
        "ordinary": [0.5, -0.25, 2.0, -3.0],
        "huge": [3e38, 3e38, -3e38, 1.0],
    }
    for name, v in cases.items():
        with np.errstate(all="ignore"):
            keys = h.hash_vector(np.array(v, dtype=np.float32)).as_tuple()
        g3.append({"name": name, "x_hex": np.array(v, dtype=np.float32).tobytes().hex(),
                   "keys_hex": [k.hex() for k in keys]})
    json.dump({"projections": [p.tolist() for p in custom], "cases": g3},
              open(os.path.join(OUT, "g3_specials.json"), "w"), indent=1)

    # ---- G4 cosine ----------------------------------------------------------
    rng = np.random.default_rng(201)
    q = rng.standard_normal(768).astype(np.float32)
    cands = rng.standard_normal((64, 768)).astype(np.float32)
    cands[5] = q * 3.0            # exact direction match
    cands[9] = -q                 # exact opposite
    cands[11] = cands[12]         # a tie pair
    scores = sim.cosine_similarity(q, cands)
    g4 = {"scores_q201_c64": scores.astype(np.float32)}
    g4j = {"topk": {}}
    for k in (1, 10, 64, 100):
        res = sim.top_k_cosine(q, cands, k=k)
        g4j["topk"][str(k)] = [[int(i), float(s)] for i, s in res]
    # literal cases of the reference's own tests (tests/test_lshrs.py:115-153)
    q3 = np.array([1.0, 0.0, 0.0], dtype=np.float32)
    c4 = [np.array(v, dtype=np.float32) for v in ([1, 0, 0], [0, 1, 0], [-1, 0, 0], [1, 1, 0])]
    g4j["ref_test_cosine"] = [float(v) for v in sim.cosine_similarity(q3, c4)]
    c5 = [np.array(v, dtype=np.float32) for v in ([1, .1, 0], [0, 1, 0], [1, 0, 0], [-1, 0, 0], [.9, .2, 0])]
    g4j["ref_test_topk3"] = [[int(i), float(s)] for i, s in sim.top_k_cosine(q3, c5, k=3)]
    g4j["ref_test_topk10_len"] = len(sim.top_k_cosine(q3, c5, k=10))
    # small-dim batch (dim=32) as LSHRS.query drives it
    rng = np.random.default_rng(202)
    corpus = rng.standard_normal((500, 32)).astype(np.float32)
    queries = rng.standard_normal((6, 32)).astype(np.float32)
    cidx = rng.integers(0, 500, size=(6, 40))
    g4["batch_scores_202"] = np.stack([sim.cosine_similarity(queries[i], corpus[cidx[i]]) for i in range(6)])
    g4j["batch_topk_202"] = [[[int(i), float(s)] for i, s in sim.top_k_cosine(queries[i], corpus[cidx[i]], k=40)]
                             for i in range(6)]
    np.savez_compressed(os.path.join(OUT, "g4_cosine.npz"), **g4)
    json.dump(g4j, open(os.path.join(OUT, "g4_cosine.json"), "w"), indent=1)

    # ---- G6 auto-config -----------------------------------------------------
    g6 = {}
    for n in (16, 32, 64, 128, 256, 512, 1024, 4096):
        for t in (0.3, 0.5, 0.7, 0.8, 0.9):
            b, r = br.get_optimal_config(n, t)
            g6[f"{n}:{t}"] = [int(b), int(r)]
    json.dump(g6, open(os.path.join(OUT, "g6_autoconfig.json"), "w"), indent=1)

    # ---- G5 orchestration (needs the full package -> in-memory redis stand-in) --
    for name in [m for m in sys.modules if m == "lshrs" or m.startswith("lshrs.")]:
        del sys.modules[name]
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "redis"))
        with open(os.path.join(td, "redis", "__init__.py"), "w") as fh:
            fh.write(
                "class ConnectionPool:\n"
                "    def __init__(self, **kw): self.kw = kw\n"
                "    def disconnect(self): pass\n"
                "class Redis:\n"
                "    def __init__(self, connection_pool=None, **kw): self.connection_pool = connection_pool\n"
            )
        sys.path.insert(0, td)
        sys.path.insert(1, REF_ROOT)
        import lshrs as ref_pkg  # noqa: F401
        from lshrs import LSHRS
        conftest = _load("ref_conftest", REF_ROOT + "/tests/conftest.py")
        MockStorage = conftest.MockStorage

        def ops_json(batch):
            return [[int(b), k.hex(), int(i)] for b, k, i in batch]

        rng = np.random.default_rng(301)
        data = rng.standard_normal((50, 32)).astype(np.float32)
        store = MockStorage()
        idx = LSHRS(dim=32, num_bands=4, rows_per_band=4, num_perm=16, buffer_size=10, seed=42,
                    storage=store, vector_fetch_fn=lambda ids: data[np.asarray(ids)])
        idx.index(list(range(50)), data)
        g5 = {"index50": {"batches": [ops_json(b) for b in store.batches]}}
        queries = data[[0, 7, 19, 33, 49]] + 0.05 * rng.standard_normal((5, 32)).astype(np.float32)
        g5["queries_hex"] = queries.astype(np.float32).tobytes().hex()
        g5["top_k_5"] = [idx.get_top_k(qv, topk=5) for qv in queries]
        g5["above_p_half"] = [[[int(i), float(s)] for i, s in idx.get_above_p(qv, p=0.5)] for qv in queries]
        g5["query_topk3_topp1"] = [[[int(i), float(s)] for i, s in idx.query(qv, top_k=3, top_p=1.0)]
                                   for qv in queries]

        # error timing: zero vector at row 7 of 12, buffer_size=10
        store2 = MockStorage()
        idx2 = LSHRS(dim=32, num_bands=4, rows_per_band=4, num_perm=16, buffer_size=10, seed=42, storage=store2)
        bad = rng.standard_normal((12, 32)).astype(np.float32)
        bad[7] = 0.0
        g5["bad_hex"] = bad.tobytes().hex()
        try:
            idx2.index(list(range(12)), bad)
            msg = None
        except ValueError as exc:
            msg = str(exc)
        g5["zero_row7"] = {"message": msg, "batches": [ops_json(b) for b in store2.batches],
                           "left_in_buffer": ops_json(idx2._buffer)}
        # negative id at row 3 of 5, big buffer
        store3 = MockStorage()
        idx3 = LSHRS(dim=32, num_bands=4, rows_per_band=4, num_perm=16, buffer_size=1000, seed=42, storage=store3)
        try:
            idx3.index([0, 1, 2, -4, 5], bad[:5])
            msg = None
        except ValueError as exc:
            msg = str(exc)
        g5["negative_row3"] = {"message": msg, "batches": [ops_json(b) for b in store3.batches],
                               "left_in_buffer": ops_json(idx3._buffer)}
        # C1-shaped plumbing numbers: 10k x 128, num_perm=64 -> (16, 4); count of ops / batches only
        store4 = MockStorage()
        idx4 = LSHRS(dim=128, num_perm=64, storage=store4, buffer_size=10_000)
        x1 = np.random.default_rng(1).standard_normal((10_000, 128)).astype(np.float32)
        idx4.index(list(range(10_000)), x1)
        h = hashlib.sha256()
        for batch in store4.batches:
            for b, k, i in batch:
                h.update(bytes([b]) + k + int(i).to_bytes(4, "little"))
        g5["c1"] = {"num_bands": idx4._hasher.num_bands, "rows_per_band": idx4._hasher.rows_per_band,
                    "batches": [len(b) for b in store4.batches], "ops_sha256": h.hexdigest(),
                    "top_k_row0": idx4.get_top_k(x1[0], topk=5)}
        json.dump(g5, open(os.path.join(OUT, "g5_orchestration.json"), "w"), indent=1)

        # ---- G8 queries at the headline shape: 3 000 clustered 768-d vectors, 100 queries ----------
        # (inputs are regenerated from the seeds below, only the reference's answers are stored)
        rng8 = np.random.default_rng(801)
        centers = rng8.standard_normal((300, 768)).astype(np.float32)
        data8 = (np.repeat(centers, 10, axis=0) + 0.3 * rng8.standard_normal((3000, 768))).astype(np.float32)
        store8 = MockStorage()
        idx8 = LSHRS(dim=768, num_perm=256, storage=store8, buffer_size=10_000, seed=42,
                     vector_fetch_fn=lambda ids: data8[np.asarray(ids)])
        idx8.index(list(range(3000)), data8)
        qrows = rng8.choice(3000, 100, replace=False)
        queries8 = (data8[qrows] + 0.05 * rng8.standard_normal((100, 768))).astype(np.float32)
        h8 = hashlib.sha256()
        for batch in store8.batches:
            for b, k, i in batch:
                h8.update(bytes([b]) + k + int(i).to_bytes(4, "little"))
        g8 = {"seed": 801, "num_bands": idx8._hasher.num_bands, "rows_per_band": idx8._hasher.rows_per_band,
              "ops_sha256": h8.hexdigest(), "query_rows": [int(v) for v in qrows],
              "top_k_10": [idx8.query(qv, top_k=10, top_p=None) for qv in queries8],
              "top_k_all": [idx8.query(qv, top_k=None, top_p=None) for qv in queries8],
              "above_p_half": [[[int(i), float(s)] for i, s in idx8.query(qv, top_k=None, top_p=0.5)] for qv in queries8],
              "topk3_topp1": [[[int(i), float(s)] for i, s in idx8.query(qv, top_k=3, top_p=1.0)] for qv in queries8]}
        json.dump(g8, open(os.path.join(OUT, "g8_queries_768.json"), "w"))

        # ---- G7 persistence: an index directory written by the reference itself ---------------
        import shutil

        g7 = os.path.join(OUT, "g7_saved_index")
        shutil.rmtree(g7, ignore_errors=True)
        idx7 = LSHRS(dim=32, num_bands=4, rows_per_band=4, num_perm=16, buffer_size=77, seed=9, storage=MockStorage(),
                     redis_password="hunter2", redis_prefix="pfx", similarity_threshold=0.6)
        idx7.save_to_disk(g7)

    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference checkout not present; fixtures can only be regenerated in the build container")
    main()
