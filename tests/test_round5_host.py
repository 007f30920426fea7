"""Round 5, host side (no GPU): the build-flag guard of the loader, the chunk plan of the pipelined pass and its struct."""

from __future__ import annotations

import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lshrs_hip.h")


def test_product_build_reports_no_measurement_switch():
    from lshrs_amd import _native

    _native.build()
    lib = _native.load()
    assert lib.lshrs_build_flags() == 0
    text = open(HEADER).read()
    assert int(re.search(r"#define\s+LSHRS_BUILD_WRONG_KEYS\s+0x(\w+)u", text).group(1), 16) == _native.BUILD_WRONG_KEYS
    assert int(re.search(r"#define\s+LSHRS_BUILD_TUNED\s+0x(\w+)u", text).group(1), 16) == _native.BUILD_TUNED


def test_loader_refuses_a_build_whose_keys_are_wrong_by_design(tmp_path):
    """an A/B build (`tools/ab_build.py -DLSHRS_AB_...`) carries the product's ABI number; it must say what
    it is (`lshrs_build_flags`) and `_native.load()` must refuse it unless LSHRS_ALLOW_AB=1."""
    lib = tmp_path / "lib_nox.so"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_build.py"), str(lib), "-DLSHRS_AB_FIX_NO_X"], check=True,
                   capture_output=True)
    probe = ("from lshrs_amd import _native\n"
             "try:\n"
             "    lib = _native.load()\n"
             "    print('LOADED', lib.lshrs_build_flags())\n"
             "except _native.NativeLibraryError as exc:\n"
             "    print('REFUSED', exc)\n")
    env = dict(os.environ, LSHRS_HIP_LIBRARY=str(lib), PYTHONPATH=ROOT)
    env.pop("LSHRS_ALLOW_AB", None)
    out = subprocess.run([sys.executable, "-c", probe], env=env, capture_output=True, text=True, check=True).stdout
    assert out.startswith("REFUSED") and "wrong by design" in out and "LSHRS_ALLOW_AB=1" in out, out
    out = subprocess.run([sys.executable, "-c", probe], env=dict(env, LSHRS_ALLOW_AB="1"), capture_output=True, text=True,
                         check=True).stdout
    flags = int(out.split()[1])
    assert out.startswith("LOADED") and flags & 1 and flags & (1 << 9), out
    # a switch that only changes speed is loaded, and says so
    lib2 = tmp_path / "lib_prio.so"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_build.py"), str(lib2), "-DLSHRS_AB_NO_STATIC_PRIO"], check=True,
                   capture_output=True)
    out = subprocess.run([sys.executable, "-c", probe], env=dict(env, LSHRS_HIP_LIBRARY=str(lib2)), capture_output=True, text=True,
                         check=True).stdout
    assert out.startswith("LOADED") and int(out.split()[1]) == 2 | (1 << 16), out


def test_struct_layouts_are_the_headers():
    """The compiler's layout of the header's structs == what the binding builds: `lshrs_sig_sort` (ctypes), and
    `lshrs_bucket_segment` (ABI 7) - uploaded as rows of six int64 by `lshrs_amd/_query_device.py`."""
    from lshrs_amd import _native

    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "lshrs_hip.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", '
           "sizeof(lshrs_bucket_segment), offsetof(lshrs_bucket_segment, codes), offsetof(lshrs_bucket_segment, offsets), "
           "offsetof(lshrs_bucket_segment, members), offsetof(lshrs_bucket_segment, n_codes) + 100 * offsetof(lshrs_bucket_segment, directory) + 10000 * offsetof(lshrs_bucket_segment, dir_codes), sizeof(lshrs_sig_sort), "
           "offsetof(lshrs_sig_sort, hist), offsetof(lshrs_sig_sort, thr), offsetof(lshrs_sig_sort, parity)); return 0; }\n")
    exe = os.path.join(ROOT, "oracle", "_build", "struct_layout")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["gcc", "-x", "c", "-", "-I" + os.path.join(ROOT, "include"), "-o", exe], input=src, text=True, check=True)
    got = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    S = _native.SigSort
    assert got == [48, 0, 8, 16, 24 + 100 * 32 + 10000 * 40, ctypes.sizeof(S), S.hist.offset, S.thr.offset, S.parity.offset]
    text = open(HEADER).read()
    assert int(re.search(r"#define\s+LSHRS_QUERY_MAX_PAIRS\s+(\d+)", text).group(1)) == _native.QUERY_MAX_PAIRS
    assert "lshrs_sig_chunk_plan" not in text and "chunked_f32" not in text          # (round 6: the chunked pass is gone)


def test_round5_experiments_are_gone_and_old_pickles_still_load():
    """no `chunking`, no `stage2_sorted="sort"`; a hasher pickled by round 5 (which carried them) loads."""
    import pickle

    from lshrs_amd import LSHHasher

    h = LSHHasher(16, 16, 768, seed=42)
    assert not hasattr(h, "chunking") and not hasattr(h, "_chunk_rows") and h.stage2_sorted == "auto" and h._stage2_mode() == 1
    h.stage2_sorted = "sort"
    with pytest.raises(ValueError, match="stage2_sorted"):
        h._stage2_mode()
    h.stage2_sorted = False
    assert h._stage2_mode() is None
    state = LSHHasher(4, 4, 32).__getstate__()
    state.update(chunking="on", chunk_min_rounds=6, _chunk_res={}, stage2_sorted="sort")
    old = LSHHasher.__new__(LSHHasher)
    old.__setstate__(pickle.loads(pickle.dumps(state)))
    assert not hasattr(old, "chunking") and old.stage2_sorted == "auto"
    # argument validation of the query entry points happens before anything touches a device
    from lshrs_amd import _native

    lib = _native.load()
    assert lib.lshrs_query_lookup_u8(None, 5, 16, 2, None, 0, None, None, None, None, None) == -10001
    assert lib.lshrs_query_lookup_u8(None, 0, 16, 2, None, 0, None, None, None, None, None) == 0
    assert lib.lshrs_query_one_u8(None, 16, 2, None, 0, None, None, None, 16384, -1, -1.0, 0, None, None, None, None, None, None, None, 0, None, None, 0, None) == -10001
    assert lib.lshrs_query_scan_i32(None, 5, -1, -1.0, None, None, None, None) == -10001
    assert lib.lshrs_query_collide_big_i64(None, 1, 16, None, None, 20000, None, None, None, None, None) == -10001
    assert lib.lshrs_query_big_workspace_bytes(20000) == 2 * 32768 * 8 + 16 and lib.lshrs_query_big_workspace_bytes(-1) < 0
    assert lib.lshrs_query_collide_pairs_i64(None, None, None, 3, 5, 16, None, None, None, None) == -10001
    assert lib.lshrs_query_rank_f32(None, None, None, None, None, None, 3, 5, None, None, None, 0, None) == -10001
    assert lib.lshrs_cosine_ragged_f32(None, 10, 8, 8, None, 3, None, None, None, 7, None, None, None) == -10001


def test_ingest_lanes_are_planned_onto_their_gpus_numa_nodes(tmp_path, monkeypatch):
    """lane <-> GPU <-> NUMA node <-> CPUs from sysfs (`lshrs_amd/numa.py`), on a fabricated 8-GPU / 2-socket
    tree; a worker thread binds itself to its node's CPUs (the ones this process may use) and back to another node's when the
    pool hands it another device; unknown topology = unbound, never an error."""
    import threading

    from lshrs_amd import numa

    root = tmp_path / "sys"
    cpus_here = sorted(os.sched_getaffinity(0))
    half = max(1, len(cpus_here) // 2)
    node_sets = [cpus_here[:half], cpus_here[half:] or cpus_here[:half]]
    for node, cpus in enumerate(node_sets):
        d = root / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(str(c) for c in cpus) + "\n")
    for card in range(8):
        bdf = f"0000:{0x10 + card:02x}:00.0"
        pci = root / "bus" / "pci" / "devices" / bdf
        pci.mkdir(parents=True)
        (pci / "vendor").write_text("0x1002\n")
        (pci / "numa_node").write_text(f"{card // 4}\n")
        drm = root / "class" / "drm" / f"card{card}"
        drm.mkdir(parents=True)
        os.symlink(pci, drm / "device")
        (root / "class" / "drm" / f"card{card}-DP-1").mkdir()
    monkeypatch.setattr(numa, "SYS_ROOT", str(root))
    plan = numa.lane_plan(range(8), use_torch=False)
    assert [e["numa_node"] for e in plan] == [0, 0, 0, 0, 1, 1, 1, 1]
    assert plan[0]["cpus"] == node_sets[0] and plan[7]["cpus"] == sorted(node_sets[1])
    folded = numa.describe_plan(plan)
    assert all(e["bound"] for e in folded) and isinstance(folded[0]["cpus"], str)
    assert numa.gpu_numa_node(8, use_torch=False) is None and numa.lane_plan([9], use_torch=False)[0]["cpus"] == []
    seen = {}

    def worker():
        seen["n0"] = numa.bind_current_thread(1, use_torch=False)
        seen["a0"] = sorted(os.sched_getaffinity(0))
        seen["n1"] = numa.bind_current_thread(6, use_torch=False)          # the same pool thread, another device
        seen["a1"] = sorted(os.sched_getaffinity(0))

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert seen["n0"] == 0 and seen["a0"] == node_sets[0] and seen["n1"] == 1 and seen["a1"] == sorted(node_sets[1])
    assert sorted(os.sched_getaffinity(0)) == cpus_here                     # the caller's own thread is never touched
    monkeypatch.setenv("LSHRS_NUMA", "0")
    assert numa.bind_current_thread(2, use_torch=False) is None
    monkeypatch.setattr(numa, "SYS_ROOT", str(tmp_path / "nothing"))
    assert numa.lane_plan([0, 1], use_torch=False) == [{"lane": 0, "device": 0, "numa_node": None, "cpus": []},
                                                        {"lane": 1, "device": 1, "numa_node": None, "cpus": []}]


def test_stage1_assembly_has_no_unpadded_accumulator_moves():
    """Round 6: sig16_kernel's matrix instructions are inline asm, hipcc pads nothing behind them.  A change of the kernel's
    template arguments once made it put accumulator moves between the drain's MFMAs and their wait states (keys of vectors with
    an odd number of k-tiles were off until the live audit refused them).  `tools/check_mfma_hazards.py` reads the unit's
    assembly (no GPU) for any v_accvgpr_* that touches an MFMA's tile too early; its scanner is checked on a made-up listing."""
    import importlib.util
    import shutil

    spec = importlib.util.spec_from_file_location("check_mfma_hazards", os.path.join(ROOT, "tools", "check_mfma_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    listing = """
_Zkernel:
	v_mfma_f32_16x16x32_bf16 a[16:19], v[14:17], v[54:57], a[16:19]
	v_accvgpr_read_b32 v61, a7
	v_accvgpr_read_b32 v69, a19
	s_nop 7
	s_nop 4
	v_accvgpr_read_b32 v70, a18
	v_mfma_f32_16x16x32_bf16 a[0:3], v[14:17], v[54:57], a[0:3]
	s_nop 7
	v_accvgpr_mov_b32 a4, a1
	s_nop 7
	v_accvgpr_write_b32 a9, 0
	s_nop 0
	v_mfma_f32_16x16x32_bf16 a[8:11], v[14:17], v[54:57], a[8:11]
	v_accvgpr_write_b32 a20, 0
	s_nop 2
	v_mfma_f32_16x16x32_bf16 a[20:23], v[14:17], v[54:57], a[20:23]
"""
    found = mod.scan(listing)
    assert [(f[1], f[4]) for f in found] == [(5, 1), (11, 8), (15, 1)], found      # (line, wait states elapsed behind the write)
    if shutil.which(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) is None:
        pytest.skip("no hipcc here")
    assert mod.main() == 0
