"""Host-resident frames through batch.remap_frames: the PCIe-bound pipeline rate on c2."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import photonbend_amd as pb
from photonbend_amd import batch
from oracle.synth import synth_frame
fov = pb.utils.to_radians(360)
dst = pb.CameraImage(np.zeros((4096, 4096, 3), np.uint8), fov, pb.equidistant(), magnitude=2047.5)
frames = [synth_frame(4096, 8192, f) for f in range(4)]
plan = batch.plan_for(dst, [], pb.PanoramaImage(frames[0]))
N = 24
list(batch.remap_frames(plan, (frames[i % 4] for i in range(4))))
t0 = time.perf_counter(); n = sum(1 for _ in batch.remap_frames(plan, (frames[i % 4] for i in range(N)))); dt = time.perf_counter() - t0
print('streamed %d host frames: %.2f ms/frame = %.0f Mpx/s (PCIe-inclusive, overlapped H2D/remap/D2H)' % (n, dt / n * 1e3, 16.777216 * n / dt))
