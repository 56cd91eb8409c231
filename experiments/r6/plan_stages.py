"""Where a warm plan preparation's HOST time goes (the diagnostic build's PB_PLAN_STAGES=1 stamps the host clock between the steps of
pb_plan_prepare_full and prints the intervals, microseconds, one line per plan on stderr).
PB_LIB_PATH=build/libphotonbend_hip_diag.so PB_PLAN_STAGES=1 python experiments/r6/plan_stages.py [case] [n]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from photonbend_amd import _native as nat
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d, rots, s = bench.build_projs(bench.CONFIGS[name])
torch.cuda.set_device(0)
ts = []
for k in range(n):
    t0 = time.perf_counter()
    p = nat.Plan(d, rots, s)
    ts.append((time.perf_counter() - t0) * 1e3)
    del p
ts = sorted(ts[2:])
print(f"{name}: warm plan creation min {ts[0]:.3f} median {ts[len(ts) // 2]:.3f} ms", file=sys.stderr)
