"""`python bench.py --gpus N` must launch its own rank processes: the driver invokes bench.py as a plain
script; with N > 1 and no launcher around it the script starts N fresh rank processes itself, before importing torch or
touching a GPU, relays rank 0's one JSON line and fails when a rank fails.  CPU only (`--dry-run`: rendezvous + one gloo
barrier, no GPU work)."""

from __future__ import annotations

import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_self_launch_two_ranks_one_json_line():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=300,
                         env=_env())
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout                      # ONE line on stdout, whatever gloo prints (it goes to stderr)
    line = json.loads(lines[0])
    # (round 6: + the lane <-> GPU <-> NUMA node plan the ranks bind to; this container has no GPUs in sysfs: every lane "unbound")
    plan = line.pop("numa_plan")
    assert line == {"dry_run": True, "n_gpus": 2, "self_launched": True}
    assert [e["lane"] for e in plan] == [0, 1] and all(set(e) == {"lane", "device", "numa_node", "cpus", "bound"} for e in plan)


def _line(text: str) -> dict:
    """The one JSON line without its NUMA plan (checked where it matters)."""
    d = json.loads([ln for ln in text.splitlines() if ln.strip().startswith("{")][0])
    d.pop("numa_plan", None)
    return d


def _free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_world_size_one_in_the_environment_still_launches_n_ranks():
    """Some schedulers export WORLD_SIZE=1 on their own: `--gpus 2` must still start two ranks (not run one silently), and a
    launcher that started another number of ranks is an error."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=300,
                         env=dict(_env(), WORLD_SIZE="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert _line(out.stdout) == {"dry_run": True, "n_gpus": 2, "self_launched": True}
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-run"], capture_output=True, text=True, timeout=120,
                         env=dict(_env(), WORLD_SIZE="2", RANK="0"))
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr and out.stdout.strip() == ""


def test_launcher_form_still_works():
    """The driver's multi-GPU form: torch.distributed.run around the same script (ranks do not self-launch again)."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "2", "--dry-run"],
                         capture_output=True, text=True, timeout=300, env=_env())
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1 and _line(lines[0]) == {"dry_run": True, "n_gpus": 2, "self_launched": False}


def test_a_failing_rank_fails_the_launch_quickly():
    """No GPU in the build container: every rank dies at `torch.cuda.set_device`; the parent must notice, stop the others
    and exit non-zero instead of waiting at a rendezvous.  (On a GPU box the two ranks would run: skipped there.)"""
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("needs a box without a GPU")
    t0 = time.time()
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--steps", "1", "--no-extras"],
                         capture_output=True, text=True, timeout=300, env=_env())
    assert out.returncode != 0 and out.stdout.strip() == "" and "rank exit codes" in out.stderr
    assert time.time() - t0 < 120


def test_eight_ranks_dry_run_under_every_world_size_the_driver_may_export():
    """the N = 8 line is the driver's to take on a whole node - what can be shown here is that `--gpus 8`
    brings up EIGHT ranks (gloo, no GPU work: `--dry-run`) whether WORLD_SIZE is unset, exported as 1 by a scheduler, or 8 with
    RANK unset (an environment left over from a launcher that is not there), and that every rank saw its own LOCAL_RANK."""
    for label, extra in (("unset", {}), ("WORLD_SIZE=1", {"WORLD_SIZE": "1"}), ("WORLD_SIZE=8 without RANK", {"WORLD_SIZE": "8"})):
        out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-run"], capture_output=True, text=True, timeout=600,
                             env=dict(_env(), **extra))
        assert out.returncode == 0, (label, out.stderr[-2000:])
        lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1 and _line(lines[0]) == {"dry_run": True, "n_gpus": 8, "self_launched": True}, (label, out.stdout)
        assert len(json.loads(lines[0])["numa_plan"]) == 8
    # under the driver's own launcher the eight ranks are the launcher's
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "8", "--dry-run"],
                         capture_output=True, text=True, timeout=600, env=_env())
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1 and _line(lines[0]) == {"dry_run": True, "n_gpus": 8, "self_launched": False}
