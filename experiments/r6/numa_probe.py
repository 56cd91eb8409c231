"""Is the single call's occasional 3-13 ms upload a NUMA effect?  Upload DMA of freshly written, freshly page-locked 100 MB arrays with the process
pinned to each NUMA node's CPUs in turn (first touch places the pages)."""
import os, sys, time, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from photonbend_amd import _device, _native as nat
torch.cuda.set_device(0)
lib = nat.load()
bdf = torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else None
print("pci bus id:", bdf, " affinity:", len(os.sched_getaffinity(0)), "cpus")
nodes = sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))
print("numa nodes:", [os.path.basename(n) for n in nodes])
for dev in glob.glob("/sys/class/drm/card*/device/numa_node") + glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")[:0]:
    try: print(dev, open(dev).read().strip())
    except Exception as ex: print(dev, ex)
def cpus_of(node):
    out = set()
    for part in open(node + "/cpulist").read().strip().split(","):
        a, _, b = part.partition("-"); out.update(range(int(a), int(b or a) + 1))
    return out
up = 100663296
din = _device.DeviceArray((up,), np.uint8)
st = _device.Stream()
allowed = os.sched_getaffinity(0)
for node in nodes + [None]:
    cp = (cpus_of(node) & allowed) if node else allowed
    if not cp: continue
    os.sched_setaffinity(0, cp)
    ts = []
    for k in range(6):
        a = np.full(up, k, np.uint8)  # (first touch here)
        nat.check(lib.pb_host_register(a.ctypes.data, a.nbytes))
        t0 = time.perf_counter(); nat.check(lib.pb_memcpy_h2d(din.data_ptr(), a.ctypes.data, up, st.handle)); st.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        nat.check(lib.pb_host_unregister(a.ctypes.data)); del a
    print(f"pinned to {os.path.basename(node) if node else 'all allowed cpus'} ({len(cp)} cpus): upload DMA ms", [round(t, 2) for t in ts], flush=True)
