import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from photonbend_amd import _native as nat
from oracle import reference_path as orc
from tests import helpers as H
from tests.cases import Case, cam, pano, dbl, inscribed
cases = [
  Case("t1", cam(33, 35, "equidistant", 180), pano(2, 4)),
  Case("t2", cam(40, 40, "equidistant", 360, inscribed(40)), pano(3, 6)),
  Case("t3", pano(5, 9), cam(3, 3, "equisolid", 180, inscribed(3)), mask=0),
  Case("t4", pano(64, 128), cam(2, 2, "equidistant", 180, inscribed(2))),
  Case("t5", pano(40, 80), dbl(2, 4, "equidistant", 190)),
  Case("t6", cam(1, 1, "equidistant", 180, 0.5), pano(8, 16)),
  Case("t7", pano(1, 2), pano(8, 16), [(10, 20, 30)]),
  Case("t8", pano(70, 140), pano(1, 2)),
  Case("t9", cam(64, 64, "rectilinear", 100, inscribed(64)), pano(2, 3)),
]
rng = np.random.default_rng(3)
for c in cases:
    try:
        h, w = c.src[1], c.src[2]
        frame = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        want = orc.remap_bilinear(H.orc_proj(c.dst), H.orc_proj(c.src), frame, H.orc_rots(c))
        plan = H.pb_plan_private(c)
        got = plan.remap(torch.from_numpy(frame).cuda(), interpolation="bilinear").cpu().numpy()
        d = np.abs(got.astype(np.int16) - want.astype(np.int16))
        if c.src[0] == "double": d = np.minimum(d, 256 - d)
        print(c.name, 'fast' if plan.info()['fast_path'] else 'f64 ', 'max diff', int(d.max()), 'n>1', int((d.max(axis=2) > 1).sum()), 'of', d.shape[0]*d.shape[1], flush=True)
    except Exception as ex:
        print(c.name, 'EXC', type(ex).__name__, str(ex)[:150], flush=True)
