"""Tile kernels against the per-pixel definition kernel on every pixel of the BASELINE geometries (noise frame): share of pixels that differ at all,
pixels beyond 1 LSB, black <-> sampled flips.   python experiments/r6/quality_census.py [case ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import full_cases
names = [a for a in sys.argv[1:] if not a.startswith('-')]
for case in full_cases():
    if names and case.name not in names:
        continue
    _, h, w, *_ = case.src
    frame = nat.synth_frame(h, w, frame=0, seed=0, circle_mask=case.mask)
    src, cmap = H.pb_chain(case, frame)
    dmap = cmap.device_tensor()
    want = nat.sample_map_bilinear(src._proj("src"), dmap, frame, 3, np.uint8).reshape(case.dst[1], case.dst[2], 3)
    del dmap
    plan = H.pb_plan_private(case)
    got = plan.remap(frame, interpolation="bilinear")
    d = (got.to(torch.int16) - want.to(torch.int16)).abs()
    if case.src[0] == "double":
        d = torch.minimum(d, 256 - d)
    ch = int((d > 0).sum())
    d = d.amax(dim=2)
    flips = (got == 0).all(dim=2) != (want == 0).all(dim=2)
    print(f"{case.name} {nat.LIB_PATH.split('/')[-1]}: pixels differing {int((d > 0).sum())} of {d.numel()} = {100.0 * int((d > 0).sum()) / d.numel():.3f} %  (channel values {100.0 * ch / (3 * d.numel()):.3f} %)  beyond 1 LSB {int((d > 1).sum())}  max {int(d.max())}  flips {int(flips.sum())}", flush=True)
    del want, got, plan, frame
    torch.cuda.empty_cache()
