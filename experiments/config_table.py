"""us/frame and algorithmic TB/s for every BASELINE config (device-resident frames, pool of 4)."""
import sys, json, torch
sys.path.insert(0, '.')
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import full_cases
pins = H.load_full()
for case in full_cases():
    plan = H.pb_plan(case)
    _, h, w, *_ = case.src
    frames = [nat.synth_frame(h, w, frame=f, circle_mask=case.mask) for f in range(4)]
    outs = [torch.empty((case.dst[1], case.dst[2], 3), dtype=torch.uint8, device='cuda') for _ in range(4)]
    for i in range(3): plan.remap(frames[i], outs[i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    N = 40
    for i in range(N): plan.remap(frames[i % 4], outs[i % 4])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / N
    alg = pins[case.name]['algorithmic_bytes']
    mpx = case.dst[1] * case.dst[2] / 1e6
    print('| %s | %.2f Mpx | %d | %.1f | %.0f | %.2f | %.1f %% |' % (case.name, mpx, alg, us, mpx / us * 1e6, alg / us / 1e6, alg / us / 1e6 / 8 * 100))
