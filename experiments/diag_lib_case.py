"""us/frame of one BASELINE case in FAST mode with an alternative build of the library: diag_lib_case.py <lib.so> <case>"""
import sys, os, torch
lib = os.path.abspath(sys.argv[1])
sys.path.insert(0, '.')
import photonbend_amd.build as b
b.LIB_PATH = lib
import photonbend_amd._native as nat
nat.LIB_PATH = lib
from tests import helpers as H
from tests.cases import full_cases
case = [c for c in full_cases() if c.name == sys.argv[2]][0]
plan = H.pb_plan(case)
_, h, w, *_ = case.src
frames = [nat.synth_frame(h, w, frame=f, circle_mask=case.mask) for f in range(4)]
outs = [torch.empty((case.dst[1], case.dst[2], 3), dtype=torch.uint8, device='cuda') for _ in range(4)]
for i in range(4): plan.remap(frames[i], outs[i])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 24
e0.record()
for i in range(N): plan.remap(frames[i % 4], outs[i % 4])
e1.record(); torch.cuda.synchronize()
print(os.path.basename(lib), case.name, '%.1f us/frame' % (e0.elapsed_time(e1) * 1e3 / N), 'budget', plan.info().get('window_budget'))
