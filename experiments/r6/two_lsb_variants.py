"""2-LSB pixels of the tile kernels under variations of one geometry (pano <- double / camera, source size, window budget)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import Case

def census(case, budget=0, frame_no=5):
    frame = nat.synth_frame(case.src[1], case.src[2], frame=frame_no)
    plan = H.pb_plan_private(case)
    if budget:
        plan.set_window_budget(budget)
    got = plan.remap(frame, interpolation="bilinear").to(torch.int16)
    src, cmap = H.pb_chain(case, frame)
    want = nat.sample_map_bilinear(src._proj("src"), cmap.device_tensor(), frame, 3, np.uint8).reshape(case.dst[1], case.dst[2], 3).to(torch.int16)
    d = (got - want).abs(); d = torch.minimum(d, 256 - d).amax(dim=2)
    mix = plan.bilinear_tile_mix()
    rows = torch.nonzero((d > 1).any(dim=1)).flatten().tolist()
    print(f"{case.name:28s} budget {budget:6d}: beyond 1: {int((d > 1).sum()):5d}  differing {100.0 * int((d > 0).sum()) / d.numel():.3f} %  of {d.numel()}  mix w{mix['window']} d{mix['direct']} t{mix['table']} td3 {mix['td3']}  rows {rows[:3]}..{rows[-3:]}", flush=True)

fov = 214.961646019831
for dh, sh in ((560, 220), (560, 440), (560, 880), (280, 220), (1120, 220), (2240, 220)):
    census(Case(f"dbl{sh}->pano{dh}", ("pano", dh, 2 * dh, "equidistant", 0.0, None), ("double", sh, 2 * sh, "equisolid", fov, None), [], 0))
    census(Case(f"cam{sh}->pano{dh}", ("pano", dh, 2 * dh, "equidistant", 0.0, None), ("camera", sh, sh, "equisolid", fov, None), [], 0))
for b in (2048, 8192, 32768):
    census(Case("dbl220->pano560", ("pano", 560, 1120, "equidistant", 0.0, None), ("double", 220, 440, "equisolid", fov, None), [], 0), budget=b)
