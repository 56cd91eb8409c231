"""Do single calls on never-seen arrays leave their registrations behind?  Prints, per call, the time, the live registrations and the process's pinned bytes."""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from photonbend_amd import _device, _hostpipe, _native as nat
torch.cuda.set_device(0)
cfg = bench.CONFIGS["c2"]
d, rots, s = bench.build_projs(cfg)
plan = nat.Plan(d, rots, s)
if os.environ.get("REG_MAX"): _device.REGISTERED._max_count = int(os.environ["REG_MAX"])
rng = np.random.default_rng(7)
pool = [rng.integers(0, 256, size=(s.height, s.width, 3), dtype=np.uint8) for _ in range(4)]
pipe = _hostpipe.pipe_for()
for k in range(10):
    a = pool[k % 4].copy()
    T = [time.perf_counter()]
    d_in = pipe.device_buffer("in", a.nbytes); d_out = pipe.device_buffer("out", 3 * plan.dst.height * plan.dst.width); T.append(time.perf_counter())
    pipe.upload(a, d_in); T.append(time.perf_counter())
    pipe.stream.sync(); T.append(time.perf_counter())
    plan.launch(d_in.data_ptr(), d_out.data_ptr(), 1, pipe.stream.handle, "nearest"); T.append(time.perf_counter())
    out = pipe.download(d_out, (plan.dst.height, plan.dst.width, 3), np.uint8); T.append(time.perf_counter())
    pipe.stream.sync(); T.append(time.perf_counter())
    names = ["buffers", "upload(issue+register)", "upload wait", "launch", "download issue (+result alloc)", "final wait"]
    print(f"call {k}: total {1e3 * (T[-1] - T[0]):.3f} ms  " + "  ".join(f"{n} {1e3 * (T[i + 1] - T[i]):.3f}" for i, n in enumerate(names)), flush=True)
