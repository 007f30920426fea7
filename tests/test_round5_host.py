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


def test_chunk_plan_struct_is_the_headers():
    from lshrs_amd import _native

    text = open(HEADER).read()
    fields = re.search(r"typedef struct lshrs_sig_chunk_plan \{(.*?)\} lshrs_sig_chunk_plan;", text, flags=re.S).group(1)
    names = re.findall(r"(\w+)(?:\[\w+\])?;", re.sub(r"/\*.*?\*/", "", fields, flags=re.S))
    assert names == [f[0] for f in _native.SigChunkPlan._fields_]
    assert int(re.search(r"#define\s+LSHRS_SIG_MAX_CHUNKS\s+(\d+)", text).group(1)) == _native.SIG_MAX_CHUNKS == 8
    # the compiler's layout of the header's struct == ctypes' layout of the binding's
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "lshrs_hip.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", '
           "sizeof(lshrs_sig_chunk_plan), offsetof(lshrs_sig_chunk_plan, rows), offsetof(lshrs_sig_chunk_plan, flag_cap), "
           "offsetof(lshrs_sig_chunk_plan, side_stream), offsetof(lshrs_sig_chunk_plan, ev_join), "
           "offsetof(lshrs_sig_chunk_plan, ev_timing)); return 0; }\n")
    exe = os.path.join(ROOT, "oracle", "_build", "chunk_plan_layout")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["gcc", "-x", "c", "-", "-I" + os.path.join(ROOT, "include"), "-o", exe], input=src, text=True, check=True)
    got = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    P = _native.SigChunkPlan
    assert got == [ctypes.sizeof(P), P.rows.offset, P.flag_cap.offset, P.side_stream.offset, P.ev_join.offset, P.ev_timing.offset]


def test_chunk_rows_plan():
    """`LSHHasher._chunk_rows`: off by default; 'on' cuts at whole rounds of stage-1 workgroups, the last chunk = the ragged
    end + one round; an explicit plan must add up."""
    from lshrs_amd import LSHHasher

    h = LSHHasher(16, 16, 768, seed=42)
    assert h.chunking == "off" and h._chunk_rows(1_000_000) is None
    h.chunking = "on"
    assert h._chunk_rows(300_000) is None                      # fewer than chunk_min_rounds rounds: one launch
    plan = h._chunk_rows(1_000_000)
    assert plan == [524_288, 393_216, 82_496] and sum(plan) == 1_000_000
    assert all(r % 65_536 == 0 for r in plan[:-1])
    h5 = LSHHasher(16, 32, 1536, seed=7)                       # two column blocks: a round is 32 768 rows
    h5.chunking = "on"
    plan = h5._chunk_rows(5_000_000)
    assert sum(plan) == 5_000_000 and all(r % 32_768 == 0 for r in plan[:-1]) and 32_768 <= plan[-1] < 2 * 32_768
    hs = LSHHasher(16, 4, 128, seed=1)                         # short vectors (resident-image kernel): never chunked
    hs.chunking = "on"
    assert hs._chunk_rows(4_000_000) is None
    h.chunking = [600_000, 400_000]
    assert h._chunk_rows(1_000_000) == [600_000, 400_000]
    assert h._chunk_rows(900_000) is None                      # a plan that does not add up is not used
    h.chunking = "sometimes"
    with pytest.raises(ValueError):
        h._chunk_rows(1_000_000)
    # argument validation of the entry point happens before anything touches a device
    from lshrs_amd import _native

    lib = _native.load()
    assert lib.lshrs_sig_hash_batch_split_replay_chunked_f32(None, 5, 32, None, 1, 1, 32, None, None, 0.0, None, None, None, 0.0,
                                                             1, None, None, None, None, None) == -10001
