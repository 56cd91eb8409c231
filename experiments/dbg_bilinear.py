import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import reference_path as orc
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.test_hip_bilinear import CASES, smooth_frame
for case in CASES:
    if case.name not in ("bl_photo_rot","bl_alter"): continue
    od, os_ = H.orc_proj(case.dst), H.orc_proj(case.src); rots = H.orc_rots(case)
    frame = smooth_frame(case.src[1], case.src[2])
    want = orc.remap_bilinear(od, os_, frame, rots)
    near = orc.remap(od, os_, frame, rots)
    plan = H.pb_plan(case)
    got = plan.remap(torch.from_numpy(frame).cuda(), interpolation="bilinear").cpu().numpy()
    plan.set_mode(nat.MODE_FAITHFUL)
    gotf = plan.remap(torch.from_numpy(frame).cuda(), interpolation="bilinear").cpu().numpy()
    d = np.abs(got.astype(np.int16) - want.astype(np.int16)).max(axis=2)
    ys, xs = np.nonzero(d > 1)
    print(case.name, plan.info())
    for y, x in zip(ys, xs):
        print(' px', y, x, 'tile', y//32, x//32, 'got', got[y,x], 'want', want[y,x], 'faithful', gotf[y,x], 'nearest', near[y,x], 'neigh want black', [(want[yy,xx]==0).all() for yy in (y-1,y,y+1) for xx in (x-1,x,x+1) if 0<=yy<want.shape[0] and 0<=xx<want.shape[1]])
