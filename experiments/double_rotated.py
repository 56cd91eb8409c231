"""Double-fisheye source chains at c5 size: plan statistics and us/frame, fast vs faithful, bytes compared."""
import sys, time, torch
sys.path.insert(0, '.')
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import Case, pano, dbl, cam, inscribed
cases = [
    Case("c5_195", pano(4096, 8192), dbl(3888, 7776, "equidistant", 195), mask=2),
    Case("c5_195_rot", pano(4096, 8192), dbl(3888, 7776, "equidistant", 195), [(3, 90, -7)], mask=2),
    Case("c5_195_rot2", pano(4096, 8192), dbl(3888, 7776, "equidistant", 195), [(30, 45, 10), (-40, 5, 77)], mask=2),
    Case("dbl_to_fisheye", cam(4096, 4096, "equidistant", 360, inscribed(4096)), dbl(3888, 7776, "equidistant", 195), [(20, 30, 40)], mask=2),
    Case("dbl_to_dbl", dbl(3888, 7776, "equisolid", 190), dbl(3888, 7776, "equidistant", 195), [(0, 15, 0)], mask=2),
]
for case in cases:
    t0 = time.perf_counter(); plan = H.pb_plan(case); torch.cuda.synchronize(); t1 = time.perf_counter()
    info = plan.info()
    print(case.name, 'plan %.1f ms' % ((t1 - t0) * 1e3), {k: info[k] for k in ('tiles', 'fix_tiles', 'fix_pixels', 'lean_tiles', 'black_tiles', 'direct_tiles')}, flush=True)
    _, h, w, *_ = case.src
    frames = [nat.synth_frame(h, w, frame=f, circle_mask=case.mask) for f in range(3)]
    outs = [torch.empty((case.dst[1], case.dst[2], 3), dtype=torch.uint8, device='cuda') for _ in range(3)]
    ref = None
    for mode, name in ((nat.MODE_FAITHFUL, 'faithful'), (nat.MODE_FAST, 'fast')):
        plan.set_mode(mode)
        for i in range(3): plan.remap(frames[i], outs[i])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        N = 12
        e0.record()
        for i in range(N): plan.remap(frames[i % 3], outs[i % 3])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / N
        if ref is None:
            ref = [o.clone() for o in outs]
            print('   %-9s %8.1f us/frame' % (name, us), flush=True)
        else:
            print('   %-9s %8.1f us/frame   differing bytes vs faithful: %d' % (name, us, sum(int((a != b).sum()) for a, b in zip(ref, outs))), flush=True)
