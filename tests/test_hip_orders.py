"""Mid-size random geometries, many with grids that divide into super-tiles, so that every launch-order rule of the plan
builder (rows from the heaviest outwards, super-tiles heaviest first, XCD exchange of double plans, plain) is exercised:
the fast path - single launches AND a batch - must reproduce the faithful float64 kernel byte for byte.  (The launch order
decides when a tile runs and on which XCD, never a pixel.)"""

import random

import numpy as np
import pytest
import torch

from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import Case, cam, dbl, inscribed, pano

pytestmark = pytest.mark.gpu

LENSES = ["equidistant", "equisolid", "stereographic", "orthographic", "thoby"]


def _cases(seed, count):
    rng = random.Random(seed)

    def side():
        return rng.choice([1024, 1280, 1536, 2048, 2304, 2560]) if rng.random() < 0.8 else rng.randrange(700, 2300)

    def fov(lens):
        return {"orthographic": rng.uniform(100, 178), "stereographic": rng.uniform(120, 300), "thoby": rng.uniform(120, 200)}.get(lens, rng.uniform(120, 360))

    def end(role):
        k = rng.random()
        if k < 0.4:
            h = side() // 2 * 2
            return pano(h, 2 * h)
        if k < 0.85 or role == "dst":
            s = side()
            lens = rng.choice(LENSES)
            return cam(s, s, lens, fov(lens), inscribed(s))
        h = side() // 2 * 2
        return dbl(h, 2 * h, "equidistant", rng.uniform(180, 200))

    out = []
    for n in range(count):
        d, s = end("dst"), end("src")
        rots = [(rng.uniform(-40, 40), rng.uniform(-180, 180), rng.uniform(-30, 30))] if rng.random() < 0.7 else []
        out.append(Case(f"ord{seed}_{n}", d, s, rots, mask=2 if s[0] == "double" else (1 if s[0] == "camera" else 0)))
    return out


CASES = _cases(7, 14)


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_fast_path_equals_faithful_whatever_the_launch_order(case):
    src, cmap = H.pb_chain(case, image=np.zeros((case.src[1], case.src[2], 3), np.uint8))
    plan = nat.Plan(cmap.dst_proj, cmap.rotations, src._proj())
    frame = nat.synth_frame(case.src[1], case.src[2], frame=5, circle_mask=case.mask)
    fast = plan.remap(frame).clone()
    batch = plan.remap(torch.stack([frame, frame]))
    assert torch.equal(batch[0], fast) and torch.equal(batch[1], fast)
    assert plan.info()["fast_path"]
    plan.set_mode(nat.MODE_FAITHFUL)
    ref = plan.remap(frame)
    assert torch.equal(fast, ref)
