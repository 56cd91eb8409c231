"""What the streamed host path's structure costs by itself (c2): the upload DMA of frame k + 1 beside the remap kernel of frame k storing into
page-locked host memory, with the real dependency (kernel k waits for upload k), rotating buffers, no host synchronisation inside the loop -
against batch.remap_frames on the same data.   python experiments/r6/stream_floor.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from photonbend_amd import _device, batch, _native as nat
from tests import helpers as H
from tests.cases import full_cases
lib = nat.load()
case = [c for c in full_cases() if c.name == "c2"][0]
plan = H.pb_plan_private(case)
up, down = 3 * case.src[1] * case.src[2], 3 * case.dst[1] * case.dst[2]
D = 3
hin = [_device.PINNED.ndarray((up,), np.uint8) for _ in range(D)]
hout = [_device.PINNED.ndarray((down,), np.uint8) for _ in range(D)]
for h in hin: h[:] = 9
din = [_device.DeviceArray((up,), np.uint8) for _ in range(D)]
s_up, s_run = _device.Stream(), _device.Stream()
ev = [_device.Event() for _ in range(D)]
done = [_device.Event() for _ in range(D)]

def run(n, dep=True, sync_each=False, same_out=False, wait_slot=True):
    t0 = time.perf_counter()
    for k in range(n):
        s = k % D
        if wait_slot and k >= D:
            done[s].sync()  # (the slot's previous kernel has read the buffer the DMA overwrites)
        nat.check(lib.pb_memcpy_h2d(din[s].data_ptr(), hin[s].ctypes.data, up, s_up.handle))
        ev[s].record(s_up)
        if dep:
            s_run.wait(ev[s])
        plan.launch(din[s].data_ptr(), hout[0 if same_out else s].ctypes.data, 1, s_run.handle, "nearest")
        done[s].record(s_run)
        if sync_each:
            ev[s].sync()
    s_up.sync(); s_run.sync()
    return (time.perf_counter() - t0) / n * 1e3

run(6)
for label, kw in (("dependent, rotating outputs, no host sync", {}), ("... host waits for each upload (iterator semantics)", {"sync_each": True}),
                  ("... one output buffer", {"same_out": True}), ("independent (kernel does not wait for the upload)", {"dep": False})):
    print("%-60s %.3f ms / frame" % (label, min(run(16, **kw) for _ in range(3))), flush=True)
rng = np.random.default_rng(5)
ring = [rng.integers(0, 256, size=(case.src[1], case.src[2], 3), dtype=np.uint8) for _ in range(4)]  # the caller's own buffers: page-locked in place when first seen
seq = [ring[k % 4] for k in range(16)]
list(batch.remap_frames(plan, seq))
for label, mk in (("batch.remap_frames, list over a ring of 4 caller buffers", lambda: seq), ("batch.remap_frames, iterator over the same", lambda: iter(seq))):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); n = sum(1 for _ in batch.remap_frames(plan, mk())); ts.append((time.perf_counter() - t0) / n * 1e3)
    print("%-60s %.3f ms / frame" % (label, min(ts)), flush=True)
