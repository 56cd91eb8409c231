"""photonbend_amd.utils.numa: the CPU-list parser and the no-op paths (no GPU, no sysfs entry) - the pinning itself needs a two-node host."""
import os

from photonbend_amd.utils import numa


def test_cpulist_parser():
    assert numa._cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert numa._cpulist("") == set()
    assert numa._cpulist("5") == {5}


def test_nothing_changes_when_the_host_does_not_say(monkeypatch):
    before = os.sched_getaffinity(0)
    monkeypatch.setattr(numa, "_pci_bus_id", lambda d: None)
    assert numa.cpus_near_device(0) is None and numa.pin_to_device(0) == 0
    monkeypatch.setattr(numa, "_pci_bus_id", lambda d: "ffff:ff:1f.0")  # no such device in sysfs
    assert numa.cpus_near_device(0) is None and numa.pin_to_device(0) == 0
    assert os.sched_getaffinity(0) == before


def test_pinning_keeps_within_the_current_affinity(monkeypatch):
    before = os.sched_getaffinity(0)
    some = set(sorted(before)[: max(1, len(before) // 2)])
    monkeypatch.setattr(numa, "cpus_near_device", lambda d=0: some | {100000})
    try:
        n = numa.pin_to_device(0)
        assert (n == len(some) and os.sched_getaffinity(0) == some) or (n == 0 and some == before)
    finally:
        os.sched_setaffinity(0, before)
