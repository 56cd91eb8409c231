import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.test_hip_bilinear import _noise_cases
for case in _noise_cases():
    frame = nat.synth_frame(case.src[1], case.src[2], frame=5)
    src, cmap = H.pb_chain(case, frame)
    want = nat.sample_map_bilinear(src._proj("src"), cmap.device_tensor(), frame, 3, np.uint8).reshape(case.dst[1], case.dst[2], 3).to(torch.int16)
    plan = H.pb_plan_private(case)
    got = plan.remap(frame, interpolation="bilinear").to(torch.int16)
    plan.set_mode(nat.MODE_FAITHFUL)
    f64 = plan.remap(frame, interpolation="bilinear").to(torch.int16)
    def dd(a,b):
        d=(a-b).abs(); return torch.minimum(d,256-d).amax(dim=2)
    d1, d2 = dd(got,want), dd(f64,want)
    print(case.name, case.src[0], 'tile vs def: >1:', int((d1>1).sum()), 'max', int(d1.max()), '| float64-mode kernel vs def: >1:', int((d2>1).sum()), 'max', int(d2.max()), '>0:', int((d2>0).sum()), flush=True)
