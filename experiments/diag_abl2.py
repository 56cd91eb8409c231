import sys, os, torch
sys.path.insert(0, '.')
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import full_cases
case = [c for c in full_cases() if c.name == sys.argv[1]][0]
plan = H.pb_plan(case)
plan.set_mode(nat.MODE_FAST)
_, h, w, *_ = case.src
frames = [nat.synth_frame(h, w, frame=f, circle_mask=case.mask) for f in range(4)]
outs = [torch.empty((case.dst[1], case.dst[2], 3), dtype=torch.uint8, device='cuda') for _ in range(4)]
for dbg in (0, 16, 4, 20, 22, 30, 31):
    os.environ['PB_DEBUG'] = str(dbg)
    for i in range(2): plan.remap(frames[i], outs[i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20): plan.remap(frames[i % 4], outs[i % 4])
    e1.record(); torch.cuda.synchronize()
    print('debug mask %2d: remap %.1f us' % (dbg, e0.elapsed_time(e1) * 1e3 / 20))
