"""PCIe-inclusive rate of the NumPy-in / NumPy-out facade on c2 (never the bench `value`)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import photonbend_amd as pb
from oracle.synth import synth_frame
fov = pb.utils.to_radians(360)
frame = synth_frame(4096, 8192, 0)
dst = pb.CameraImage(np.zeros((4096, 4096, 3), np.uint8), fov, pb.equidistant(), magnitude=2047.5)
src = pb.PanoramaImage(frame)
out = src.process_coordinate_map(dst.get_coordinate_map())   # warm-up: plan creation
ts = []
for _ in range(5):
    t0 = time.perf_counter(); out = src.process_coordinate_map(dst.get_coordinate_map()); ts.append(time.perf_counter() - t0)
t = sorted(ts)[len(ts) // 2]
print('facade numpy->numpy c2: %.2f ms/frame = %.0f Mpx/s (H2D 100.7 MB pageable + kernel + D2H 50.3 MB)' % (t * 1e3, 16.777216 / t))
pin = torch.from_numpy(frame).pin_memory(); dev = torch.empty_like(pin, device='cuda'); o = torch.empty((4096, 4096, 3), dtype=torch.uint8, device='cuda'); hp = torch.empty((4096, 4096, 3), dtype=torch.uint8).pin_memory()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): dev.copy_(pin, non_blocking=True); torch.cuda.synchronize()
h2d = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
for _ in range(5): hp.copy_(o, non_blocking=True); torch.cuda.synchronize()
d2h = (time.perf_counter() - t0) / 5
print('pinned H2D %.2f ms (%.1f GB/s), D2H %.2f ms (%.1f GB/s)' % (h2d * 1e3, 100.66e-3 / h2d / 1e0 * 1e0, d2h * 1e3, 50.33e-3 / d2h))
