"""N launches of one BASELINE config (for rocprofv3): python run_case.py c5_180 [N]"""
import sys, torch
sys.path.insert(0, '/root/repo' if False else '.')
import os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import full_cases
case = [c for c in full_cases() if c.name == sys.argv[1]][0]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
plan = H.pb_plan(case)
_, h, w, *_ = case.src
frames = [nat.synth_frame(h, w, frame=f, circle_mask=case.mask) for f in range(3)]
outs = [torch.empty((case.dst[1], case.dst[2], 3), dtype=torch.uint8, device='cuda') for _ in range(3)]
for i in range(N): plan.remap(frames[i % 3], outs[i % 3])
torch.cuda.synchronize()
