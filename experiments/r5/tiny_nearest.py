import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import reference_path as orc
from tests import helpers as H
from tests.test_hip_bilinear import _TINY
rng = np.random.default_rng(3)
for c in _TINY:
    try:
        frame = rng.integers(0, 256, size=(c.src[1], c.src[2], 3), dtype=np.uint8)
        want = orc.remap(H.orc_proj(c.dst), H.orc_proj(c.src), frame, H.orc_rots(c))
        plan = H.pb_plan_private(c)
        got = plan.remap(torch.from_numpy(frame).cuda()).cpu().numpy()
        print(c.name, 'equal' if np.array_equal(got, want) else f'DIFF {int((got != want).any(axis=2).sum())} px', flush=True)
    except Exception as ex:
        print(c.name, 'EXC', type(ex).__name__, str(ex)[:200], flush=True)
