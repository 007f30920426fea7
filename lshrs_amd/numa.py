"""Where the ingest lanes run: every lane's thread - and the pinned blocks it allocates and touches first - on the NUMA node
its GPU hangs off (round 6; SURVEY §8e: "one host thread + one stream per device; double-buffered pinned H2D").

Host-fed ingestion is bound by the PCIe link and by the host's memory system: a lane whose staging copies cross the
socket interconnect shares that link with seven others.  On an 8 x MI355X node the GPUs hang off two sockets (four each);
``/sys/bus/pci/devices/<bdf>/numa_node`` says which.  Nothing here is required for correctness: where the topology cannot be
read (a container without sysfs, a VM that reports -1) every function degrades to "unbound".
"""

from __future__ import annotations

import glob
import os
import threading
from typing import Dict, List, Optional, Sequence, Set

__all__ = ["gpu_numa_node", "node_cpus", "lane_plan", "bind_current_thread", "describe_plan"]

_local = threading.local()
SYS_ROOT = "/sys"          # (tests point this at a fabricated tree)


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as fh:
            return fh.read().strip()
    except OSError:
        return None


def _parse_cpulist(text: str) -> Set[int]:
    cpus: Set[int] = set()
    for part in text.split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def node_cpus(node: int) -> Set[int]:
    """CPUs of a NUMA node (``/sys/devices/system/node/node<N>/cpulist``), empty when unknown."""
    text = _read(f"{SYS_ROOT}/devices/system/node/node{int(node)}/cpulist")
    return _parse_cpulist(text) if text else set()


def _amd_gpu_bdfs() -> List[str]:
    """PCI addresses of the AMD display-class devices the kernel knows, in DRM card order (the order HIP enumerates a node's
    GPUs in when no ``*_VISIBLE_DEVICES`` re-orders them)."""
    cards = []
    for dev in glob.glob(f"{SYS_ROOT}/class/drm/card[0-9]*/device"):
        name = os.path.basename(os.path.dirname(dev))
        if "-" in name:                        # card0-DP-1 ...: connectors
            continue
        if (_read(os.path.join(dev, "vendor")) or "").lower() != "0x1002":
            continue
        try:
            cards.append((int(name[4:]), os.path.basename(os.path.realpath(dev))))
        except ValueError:
            continue
    return [bdf for _, bdf in sorted(cards)]


def gpu_numa_node(index: int, *, use_torch: bool = True) -> Optional[int]:
    """NUMA node of GPU ``index`` (HIP device ordinal), or None when it cannot be told.  With an initialised torch the
    device's own PCI address is asked for (``*_VISIBLE_DEVICES`` respected); else DRM card order."""
    bdf = None
    if use_torch:
        try:
            import torch

            if torch.cuda.is_available() and index < torch.cuda.device_count():
                p = torch.cuda.get_device_properties(index)
                if hasattr(p, "pci_bus_id"):
                    bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{getattr(p, 'pci_device_id', 0):02x}.0"
        except Exception:          # noqa: BLE001 - topology is advice, never an error
            bdf = None
    if bdf is None or not os.path.exists(f"{SYS_ROOT}/bus/pci/devices/{bdf}"):
        bdfs = _amd_gpu_bdfs()
        bdf = bdfs[index] if 0 <= index < len(bdfs) else None
    if bdf is None:
        return None
    text = _read(f"{SYS_ROOT}/bus/pci/devices/{bdf}/numa_node")
    try:
        node = int(text) if text is not None else -1
    except ValueError:
        node = -1
    return node if node >= 0 else None


def lane_plan(devices: Sequence[int], *, use_torch: bool = True) -> List[Dict[str, object]]:
    """One entry per ingest lane: ``{"lane", "device", "numa_node", "cpus"}`` - ``cpus`` the CPUs of that node this process may
    run on (empty: unknown or none allowed - the lane stays unbound)."""
    try:
        allowed = set(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        allowed = set(range(os.cpu_count() or 1))
    plan = []
    for lane, dev in enumerate(devices):
        node = gpu_numa_node(int(dev), use_torch=use_torch)
        cpus = sorted(node_cpus(node) & allowed) if node is not None else []
        plan.append({"lane": lane, "device": int(dev), "numa_node": node, "cpus": cpus})
    return plan


def describe_plan(plan: List[Dict[str, object]]) -> List[Dict[str, object]]:
    """The plan with the CPU lists folded to ranges (for a JSON line)."""
    def fold(cpus):
        out, run = [], []
        for c in list(cpus) + [None]:
            if run and (c is None or c != run[-1] + 1):
                out.append(f"{run[0]}-{run[-1]}" if len(run) > 1 else str(run[0]))
                run = []
            if c is not None:
                run.append(c)
        return ",".join(out)

    return [{"lane": e["lane"], "device": e["device"], "numa_node": e["numa_node"], "cpus": fold(e["cpus"]),
             "bound": bool(e["cpus"])} for e in plan]


def bind_current_thread(device: int, *, use_torch: bool = True) -> Optional[int]:
    """Pin the CALLING thread to the CPUs of ``device``'s NUMA node (once per thread; a worker thread of an ingest lane - never
    the caller's own thread, whose affinity is the caller's business).  Memory the thread allocates and touches from here on
    - the pinned staging blocks of its lane - is placed on that node by the kernel's default local policy.  Returns the node,
    or None when nothing was changed.  ``LSHRS_NUMA=0`` turns it off."""
    if os.environ.get("LSHRS_NUMA", "1") == "0":
        return None
    done = getattr(_local, "bound", None)
    if done is not None and done[0] == device:
        return done[1]
    node = gpu_numa_node(device, use_torch=use_torch)
    bound = None
    if node is not None:
        try:
            base = getattr(_local, "base", None)
            if base is None:
                base = _local.base = set(os.sched_getaffinity(0))      # what the thread was allowed before any lane bound it
            cpus = node_cpus(node) & base
            if cpus:
                os.sched_setaffinity(0, cpus)       # (pid 0 = the calling THREAD on Linux)
                bound = node
        except (AttributeError, OSError):
            bound = None
    _local.bound = (device, bound)
    return bound
