"""Plan preparation under rocprofv3 --hip-trace: N warm preparations of one geometry (the HIP API calls' count and time tell where the host time goes)."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import torch
import bench
from photonbend_amd import _native as nat
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = bench.CONFIGS[name]
d, rots, s = bench.build_projs(cfg)
torch.cuda.set_device(0)
for _ in range(3):
    p = nat.Plan(d, rots, s, budget=7168); del p
torch.cuda.synchronize()
ts = []
for _ in range(n):
    t0 = time.perf_counter(); p = nat.Plan(d, rots, s, budget=7168); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3); del p
ts.sort()
print(f"{name}: warm plan preparation min {ts[0]:.3f} median {ts[len(ts) // 2]:.3f} ms over {n}")
