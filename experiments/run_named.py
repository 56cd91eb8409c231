"""N launches of a named case from experiments/double_rotated.py or tests.cases (for rocprofv3): run_named.py <name> [N]"""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import Case, pano, dbl, cam, inscribed, full_cases
extra = [
    Case("c5_195_rot", pano(4096, 8192), dbl(3888, 7776, "equidistant", 195), [(3, 90, -7)], mask=2),
    Case("dbl_to_fisheye", cam(4096, 4096, "equidistant", 360, inscribed(4096)), dbl(3888, 7776, "equidistant", 195), [(20, 30, 40)], mask=2),
]
case = [c for c in full_cases() + extra if c.name == sys.argv[1]][0]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
plan = H.pb_plan(case)
print(plan.info())
_, h, w, *_ = case.src
frames = [nat.synth_frame(h, w, frame=f, circle_mask=case.mask) for f in range(3)]
outs = [torch.empty((case.dst[1], case.dst[2], 3), dtype=torch.uint8, device='cuda') for _ in range(3)]
for i in range(N): plan.remap(frames[i % 3], outs[i % 3])
torch.cuda.synchronize()
