"""20 warm plan preparations of one config (for rocprofv3 --hip-trace --stats): python experiments/r4/plan_only.py c3"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from photonbend_amd import _native as nat
cfg = bench.CONFIGS[sys.argv[1]]
d, rots, s = bench.build_projs(cfg)
keep = nat.Plan(d, rots, s)
torch.cuda.synchronize()
ts = []
for _ in range(20):
    t0 = time.perf_counter()
    p = nat.Plan(d, rots, s)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
    del p
print(sys.argv[1], "warm plan ms: min %.3f median %.3f" % (min(ts), sorted(ts)[10]))
