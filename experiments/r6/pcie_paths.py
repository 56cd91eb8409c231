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
