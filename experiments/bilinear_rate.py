import sys, torch
sys.path.insert(0, '.')
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import full_cases
for name in ('c2', 'c3', 'c1'):
    case = [c for c in full_cases() if c.name == name][0]
    plan = H.pb_plan(case)
    _, h, w, *_ = case.src
    frames = [nat.synth_frame(h, w, frame=f, circle_mask=case.mask) for f in range(4)]
    outs = [torch.empty((case.dst[1], case.dst[2], 3), dtype=torch.uint8, device='cuda') for _ in range(4)]
    for i in range(3): plan.remap(frames[i], outs[i], interpolation='bilinear')
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20): plan.remap(frames[i % 4], outs[i % 4], interpolation='bilinear')
    e1.record(); torch.cuda.synchronize()
    print('%s bilinear: %.1f us/frame' % (name, e0.elapsed_time(e1) * 1e3 / 20))
