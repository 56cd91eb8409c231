"""How a c2 frame pair crosses PCIe on this box (100.7 MB up, 50.3 MB down, page-locked host memory): the two DMA directions alone and together,
a device COPY KERNEL storing into page-locked host memory (16 B per lane) alone and beside the upload DMA, and the remap kernel storing its
output straight into page-locked host memory.   python experiments/r6/pcie_paths.py"""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from photonbend_amd import _device, _native as nat
from tests import helpers as H
from tests.cases import full_cases

lib = nat.load()
up, down = 100663296, 50331648
hin, hout = _device.PINNED.ndarray((up,), np.uint8), _device.PINNED.ndarray((down,), np.uint8)
hin[:] = 7
din, dout = _device.DeviceArray((up,), np.uint8), _device.DeviceArray((down,), np.uint8)
s1, s2 = _device.Stream(), _device.Stream()

def t(fn, n=8):
    fn(); s1.sync(); s2.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    s1.sync(); s2.sync()
    return (time.perf_counter() - t0) / n * 1e3

h2d = lambda s: nat.check(lib.pb_memcpy_h2d(din.data_ptr(), hin.ctypes.data, up, s.handle))
d2h = lambda s: nat.check(lib.pb_memcpy_d2h(hout.ctypes.data, dout.data_ptr(), down, s.handle))
kcopy = lambda s: nat.check(lib.pb_stream_copy(hout.ctypes.data, dout.data_ptr(), down, s.handle))   # device kernel -> host memory
print("h2d DMA alone                 %.3f ms" % t(lambda: h2d(s1)))
print("d2h DMA alone                 %.3f ms" % t(lambda: d2h(s2)))
print("h2d DMA + d2h DMA, 2 streams  %.3f ms" % t(lambda: (h2d(s1), d2h(s2))))
try:
    print("copy kernel -> host alone     %.3f ms" % t(lambda: kcopy(s2)))
    print("h2d DMA + copy kernel -> host %.3f ms" % t(lambda: (h2d(s1), kcopy(s2))))
except Exception as exc:
    print("copy kernel into host memory:", exc)
case = [c for c in full_cases() if c.name == "c2"][0]
plan = H.pb_plan_private(case)
frame = nat.synth_frame(case.src[1], case.src[2], frame=0)
try:
    launch = lambda s: plan.launch(frame.data_ptr(), hout.ctypes.data, 1, s.handle, "nearest")
    print("remap kernel -> host alone    %.3f ms" % t(lambda: launch(s2)))
    print("h2d DMA + remap kernel -> host %.3f ms" % t(lambda: (h2d(s1), launch(s2))))
    import torch
    want = plan.remap(frame).cpu().numpy()
    print("remap into host memory is the device result:", bool(np.array_equal(hout.reshape(want.shape), want)))
except Exception as exc:
    print("remap kernel into host memory:", exc)

# ---- what an ndarray the library has never seen costs to get across: page-lock it in place (hipHostRegister) + one DMA, against the chunked staging copy
from photonbend_amd import _hostpipe
pipe = _hostpipe.pipe_for()
rng = np.random.default_rng(3)
arrs = [rng.integers(0, 256, size=(4096, 8192, 3), dtype=np.uint8) for _ in range(6)]
ts = []
for a in arrs[:3]:
    t0 = time.perf_counter(); nat.check(lib.pb_host_register(a.ctypes.data, a.nbytes)); t1 = time.perf_counter()
    nat.check(lib.pb_memcpy_h2d(din.data_ptr(), a.ctypes.data, up, s1.handle)); s1.sync(); t2 = time.perf_counter()
    nat.check(lib.pb_host_unregister(a.ctypes.data)); t3 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
print("register / DMA / unregister of a fresh 100.7 MB ndarray (ms):", [tuple(round(x, 3) for x in t) for t in ts])
ts = []
for a in arrs[3:]:
    t0 = time.perf_counter(); pipe.upload(a, din, s1); s1.sync(); ts.append((time.perf_counter() - t0) * 1e3)
print("chunked staged upload of a fresh 100.7 MB ndarray (ms):", [round(x, 3) for x in ts])

# ---- does page-locking wait for the device?  (register a fresh frame while a kernel stores over PCIe / while an upload DMA runs)
more = [rng.integers(0, 256, size=(4096, 8192, 3), dtype=np.uint8) for _ in range(4)]
for label, busy in (("idle", lambda: None), ("while the remap kernel stores into host memory", lambda: launch(s2)), ("while an upload DMA runs", lambda: h2d(s1))):
    a = more.pop()
    busy()
    t0 = time.perf_counter(); nat.check(lib.pb_host_register(a.ctypes.data, a.nbytes)); t1 = time.perf_counter()
    s1.sync(); s2.sync()
    nat.check(lib.pb_host_unregister(a.ctypes.data))
    print("register a fresh 100.7 MB ndarray, device %s: %.3f ms" % (label, (t1 - t0) * 1e3))
