"""Host placement for the NumPy-in / NumPy-out path: the CPUs of the NUMA node a GPU hangs off.

Frames the caller owns are page-locked in place and read by the upload DMA where they lie: on the two-socket hosts of the MI355X pool a
100.7 MB upload out of the GPU's own node takes 1.78 ms, out of the other node 1.90-1.95 (experiments/r6/numa_probe.py), and the streamed
path is bound by exactly that DMA.  First touch places pages, so it is enough that the process (one per GPU) RUNS on the right node while
it fills its buffers: ``pin_to_device`` restricts the calling process's CPU affinity to that node (what ``numactl --cpunodebind`` would
do from outside).  Opt-in: a library does not move its host process uninvited; ``bench.py`` calls it for every rank."""

from __future__ import annotations

import os


def _pci_bus_id(device: int):
    try:
        import torch

        p = torch.cuda.get_device_properties(device)
        return "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


def _cpulist(text: str) -> set:
    out = set()
    for part in text.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            out.update(range(int(a), int(b or a) + 1))
    return out


def cpus_near_device(device: int = 0):
    """The CPUs of the NUMA node of GPU `device` (a set), or None when the host does not say (one node, no sysfs, no PyTorch to ask for
    the PCI address)."""
    bdf = _pci_bus_id(device)
    if bdf is None:
        return None
    try:
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        return _cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read()) or None
    except (OSError, ValueError):
        return None


def pin_to_device(device: int = 0) -> int:
    """Restricts this process to the CPUs of `device`'s NUMA node (within its current affinity).  Returns the number of CPUs it now runs
    on, 0 when nothing was changed."""
    near = cpus_near_device(device)
    if not near or not hasattr(os, "sched_setaffinity"):
        return 0
    keep = near & os.sched_getaffinity(0)
    if not keep or keep == os.sched_getaffinity(0):
        return 0
    os.sched_setaffinity(0, keep)
    return len(keep)
