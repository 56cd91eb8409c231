"""Do an upload and a download on two streams overlap on this box?  (host_path streams H2D of frame k + 1 against D2H of frame k - 1.)
100.7 MB up, 50.3 MB down (c2's frame pair), page-locked host memory, pb_memcpy_h2d / pb_memcpy_d2h (hipMemcpyAsync)."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from photonbend_amd import _device, _native as nat

lib = nat.load()
up, down = 100663296, 50331648
hin, hout = _device.PINNED.ndarray((up,), np.uint8), _device.PINNED.ndarray((down,), np.uint8)
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
print("h2d alone      %.3f ms" % t(lambda: h2d(s1)))
print("d2h alone      %.3f ms" % t(lambda: d2h(s2)))
print("both, 2 streams %.3f ms" % t(lambda: (h2d(s1), d2h(s2))))
print("both, 1 stream  %.3f ms" % t(lambda: (h2d(s1), d2h(s1))))
