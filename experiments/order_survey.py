"""Launch duration of a spread of geometries under the current launch-order policy (run twice: PB_ORDER=1 and default).
    python experiments/order_survey.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import photonbend_amd._native as nat
from tests import helpers as H
from tests.cases import Case, cam, dbl, pano, inscribed, full_frame
CASES = [
    Case("pano_to_pano_rot", pano(2048, 4096), pano(2048, 4096), [(20, 40, 10)]),
    Case("pano_to_pano_8k", pano(4096, 8192), pano(4096, 8192), [(5, 170, -3)]),
    Case("rectilinear_from_pano", cam(2048, 3072, "rectilinear", 120, full_frame(2048, 3072)), pano(4096, 8192), [(10, 30, 0)]),
    Case("stereographic_from_pano", cam(4096, 4096, "stereographic", 300, inscribed(4096)), pano(4096, 8192)),
    Case("pano_from_equisolid", pano(2048, 4096), cam(4096, 4096, "equisolid", 200, inscribed(4096)), [(0, 0, 30)], mask=1),
    Case("fisheye_crop_from_pano", cam(3072, 4096, "equidistant", 220, 2200.0), pano(4096, 8192), [(15, 0, 0)]),
    Case("fisheye_to_fisheye_small_rot", cam(4096, 4096, "equidistant", 360, inscribed(4096)), cam(4096, 4096, "equisolid", 360, inscribed(4096)), [(3, 2, 1)], mask=1),
    Case("double_to_pano_195_rot", pano(4096, 8192), dbl(3888, 7776, "equidistant", 195), [(2, 5, -1)], mask=2),
    Case("double_to_fisheye", cam(4096, 4096, "equidistant", 360, inscribed(4096)), dbl(3888, 7776, "equidistant", 190), mask=2),
]
import os
sel=os.environ.get('SURVEY_ONLY')
for case in CASES:
    if sel and case.name not in sel.split(','): continue
    src, cmap = H.pb_chain(case, image=np.zeros((case.src[1], case.src[2], 3), np.uint8))
    plan = nat.Plan(cmap.dst_proj, cmap.rotations, src._proj())
    _, h, w, *_ = case.src
    n = 5
    frames = [nat.synth_frame(h, w, frame=f, circle_mask=case.mask) for f in range(n)]
    outs = [torch.empty((case.dst[1], case.dst[2], 3), dtype=torch.uint8, device='cuda') for _ in range(n)]
    for k in range(2 * n): plan.remap(frames[k % n], outs[k % n])
    torch.cuda.synchronize()
    ts = []
    for rep in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(2 * n): plan.remap(frames[k % n], outs[k % n])
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / (2 * n))
    info = plan.info()
    print('%-30s %7.2f us  (lean %d direct %d black %d fail %d of %d)' % (case.name, np.median(ts), info['lean_tiles'], info['direct_tiles'], info['black_tiles'], info['fix_tiles'], info['tiles']), flush=True)
    del plan, frames, outs
    torch.cuda.empty_cache()
