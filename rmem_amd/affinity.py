"""CPU placement of the ranks of one node: each rank's host threads on the NUMA node its GPU hangs off.

The reference starts one worker process per GPU (aot_plus/tools/eval.py:137-143) and leaves their placement to the
kernel.  Here a rank runs three host threads that matter (the engine thread, the graph-launch helper of the encoder
prefetch, ROCr's event thread: ~1.1 busy cores per rank); on a two-socket host with eight GPUs
(profiles/r04_numa_topo.txt: GPUs 0-2, 7 on node 0, GPUs 3-6 on node 1) a rank that wanders to the other socket pays a
cross-socket hop on every doorbell write and pinned-memory read.  `pin_rank` restricts the process to an even share of
its GPU's node -- physical cores and their SMT siblings together -- computed from sysfs only, the same on every rank
without communication; where the topology cannot be read it does nothing.  RMEM_PIN=0 switches it off.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (sysfs cpulist format)."""
    out: List[int] = []
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def gpu_numa_node(device_index: int) -> Optional[int]:
    """NUMA node of a GPU from its PCI address (sysfs); None when it cannot be determined."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:
        return None
    v = _read(f"/sys/bus/pci/devices/{bdf}/numa_node")
    if v is None:
        return None
    try:
        n = int(v)
    except ValueError:
        return None
    return n if n >= 0 else None


def node_cpus(node: int) -> List[int]:
    return parse_cpulist(_read(f"/sys/devices/system/node/node{node}/cpulist") or "")


def smt_groups(cpus: List[int]) -> List[List[int]]:
    """The CPUs grouped by physical core (thread siblings together), in order of each core's first CPU."""
    seen, groups = set(), []
    allowed = set(cpus)
    for c in sorted(cpus):
        if c in seen:
            continue
        sib = [s for s in parse_cpulist(_read(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") or str(c))
               if s in allowed] or [c]
        seen.update(sib)
        groups.append(sorted(sib))
    return groups


def plan_node_shares(rank_nodes: List[Optional[int]], cpus_of_node: Dict[int, List[List[int]]]) -> List[Optional[List[int]]]:
    """Pure planning step (tested on the CPU): rank_nodes[r] = NUMA node of rank r's GPU (None = unknown),
    cpus_of_node[n] = that node's physical cores as sibling groups.  The ranks of a node split its cores into contiguous,
    equal shares in rank order; a rank with an unknown node, or a node with fewer cores than ranks, gets None (no pin)."""
    out: List[Optional[List[int]]] = [None] * len(rank_nodes)
    by_node: Dict[int, List[int]] = {}
    for r, n in enumerate(rank_nodes):
        if n is not None:
            by_node.setdefault(n, []).append(r)
    for n, ranks in by_node.items():
        cores = cpus_of_node.get(n) or []
        if len(cores) < len(ranks):
            continue
        per = len(cores) // len(ranks)
        for j, r in enumerate(ranks):
            out[r] = sorted(c for g in cores[j * per:(j + 1) * per] for c in g)
    return out


def pin_rank(local_rank: int, local_world: int, device_of_rank) -> Dict[str, object]:
    """Pin this process (and the threads it starts later) to its share of its GPU's NUMA node.  device_of_rank(r) ->
    device index of local rank r.  Returns what was done, for the bench line."""
    info: Dict[str, object] = {"pinned": False}
    if os.environ.get("RMEM_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        info["why"] = "disabled" if os.environ.get("RMEM_PIN", "1") == "0" else "no sched_setaffinity"
        return info
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except OSError:
        return info
    nodes = [gpu_numa_node(device_of_rank(r)) for r in range(local_world)]
    info["numa_node"] = nodes[local_rank] if local_rank < len(nodes) else None
    cpus_of_node = {}
    for n in set(x for x in nodes if x is not None):
        cpus_of_node[n] = smt_groups([c for c in node_cpus(n) if c in set(allowed)])
    share = plan_node_shares(nodes, cpus_of_node)[local_rank] if local_rank < len(nodes) else None
    if not share:
        info["why"] = "topology not readable or too few CPUs on the node"
        info["cpus_allowed"] = len(allowed)
        return info
    try:
        os.sched_setaffinity(0, share)
    except OSError as e:
        info["why"] = f"sched_setaffinity: {e}"
        return info
    # sched_setaffinity(0, ...) moves the CALLING thread only; threads that already exist -- finding the GPU's PCI address
    # above initialises the HIP runtime, whose helper threads (ROCr's event loop among them) are running by now -- keep
    # the old mask, and only threads created from here on inherit the new one.  Move every existing thread too.
    moved = 0
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), share)
                moved += 1
            except (OSError, ValueError):
                pass
    except OSError:
        pass
    info.update(pinned=True, cpus=len(share), first_cpu=share[0], last_cpu=share[-1], threads_moved=moved)
    return info
