"""Where a streamed frame's milliseconds go (batch.remap_frames, c2, ndarrays the library has never seen vs a ring of caller buffers): wall time per
frame and the host-side time inside REGISTERED.is_registered / the upload call / the wait for the upload / the launch."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import bench
from photonbend_amd import _device, _hostpipe, batch, _native as nat
cfg = bench.CONFIGS["c2"]
d, rots, s = bench.build_projs(cfg)
plan = nat.Plan(d, rots, s)
sh = (s.height, s.width, 3)
rng = np.random.default_rng(7)
pool = [rng.integers(0, 256, size=sh, dtype=np.uint8) for _ in range(4)]
acc = {}
def timed(obj, name, label):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, w)
timed(_device.REGISTERED, "is_registered", "is_registered")
timed(_device.Event, "sync", "event.sync")
timed(nat.Plan, "launch", "launch")
timed(_device._PinnedPool, "ndarray", "pinned.ndarray")
timed(_hostpipe.HostPipe, "upload", "pipe.upload (incl. is_registered)")
timed(_device.Event, "record", "event.record")
timed(_device.Stream, "wait", "stream.wait")
list(batch.remap_frames(plan, (pool[k % 4] for k in range(8))))
for label, frames in (("ring of 4 caller buffers", lambda n: (pool[k % 4] for k in range(n))), ("never-seen arrays (a list)", lambda n: [pool[k % 4].copy() for k in range(n)]), ("never-seen arrays (an iterator)", lambda n: iter([pool[k % 4].copy() for k in range(n)]))):
    for rep in range(2):
        n = 16
        it = frames(n)  # (made before the clock starts)
        acc.clear()
        t0 = time.perf_counter()
        cnt = sum(1 for _ in batch.remap_frames(plan, it))
        dt = (time.perf_counter() - t0) / n * 1e3
        print(f"{label}: {dt:.3f} ms per frame; host time per frame inside: " + ", ".join(f"{k} {v / n * 1e3:.3f}" for k, v in sorted(acc.items())), flush=True)
