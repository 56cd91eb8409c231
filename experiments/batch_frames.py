"""Frames per launch: pb_remap_u8(n_frames = B) reuses the per-pixel index math across the batch."""
import sys, torch
sys.path.insert(0, '.')
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import full_cases
case = [c for c in full_cases() if c.name == sys.argv[1]][0]
plan = H.pb_plan(case)
_, h, w, *_ = case.src
for B in (1, 2, 4, 8):
    pools = [torch.stack([nat.synth_frame(h, w, frame=10 * p + f, circle_mask=case.mask) for f in range(B)]) for p in range(max(2, 8 // B))]
    outs = [torch.empty((B, case.dst[1], case.dst[2], 3), dtype=torch.uint8, device='cuda') for _ in pools]
    for i in range(2): plan.remap(pools[i % len(pools)], outs[i % len(pools)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = max(4, 40 // B)
    e0.record()
    for i in range(N): plan.remap(pools[i % len(pools)], outs[i % len(pools)])
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e3 / (N * B)
    print('%s: %d frame(s) per launch: %.1f us per frame  (%.0f Mpx/s)' % (case.name, B, t, case.dst[1] * case.dst[2] / t))
