"""How far are the bilinear TILE kernels from the definition at full size?  (tests/test_hip_bilinear.py's allowances: are they used?)
Per BASELINE geometry, noise frame: the tile kernels against the per-pixel float64 definition kernel of the same library (MODE_FAITHFUL),
every pixel: histogram of |difference| (mod 256 for the double blend), black <-> sampled flips."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import full_cases
for case in full_cases():
    plan = H.pb_plan_private(case)
    frame = nat.synth_frame(case.src[1], case.src[2], frame=0, seed=0, circle_mask=case.mask)
    got = plan.remap(frame, interpolation="bilinear").to(torch.int16)
    plan.set_mode(nat.MODE_FAITHFUL)
    want = plan.remap(frame, interpolation="bilinear").to(torch.int16)
    d = (got - want).abs()
    if case.src[0] == "double":
        d = torch.minimum(d, 256 - d)
    d = d.amax(dim=2)
    gb, wb = (got == 0).all(dim=2), (want == 0).all(dim=2)
    flips = gb != wb
    n = d.numel()
    hist = [int((d == k).sum()) for k in range(4)]
    print(case.name, "pixels", n, "diff 0/1/2/3:", hist, ">3:", int((d > 3).sum()), "flips", int(flips.sum()), "max off-flip", int(d[~flips].max()), flush=True)
