import sys, numpy as np, torch
sys.path.insert(0, '.')
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.test_hip_bilinear import CASES, smooth_frame
from oracle import reference_path as orc
name = sys.argv[1]
case = [c for c in CASES if c.name == name][0]
frame = smooth_frame(case.src[1], case.src[2])
plan = H.pb_plan(case)
dev = torch.from_numpy(frame).cuda()
got = plan.remap(dev, interpolation='bilinear').cpu().numpy().astype(np.int16)
plan.set_mode(nat.MODE_FAITHFUL)
f64 = plan.remap(dev, interpolation='bilinear').cpu().numpy().astype(np.int16)
plan.set_mode(nat.MODE_AUTO)
want = orc.remap_bilinear(H.orc_proj(case.dst), H.orc_proj(case.src), frame, H.orc_rots(case)).astype(np.int16)
idx, w = plan.index_map(weights=True)
idx, w = idx.cpu().numpy(), w.cpu().numpy()
d = np.abs(got - f64).max(axis=2); d = np.minimum(d, 256 - d)
print('tiles vs f64: >2 LSB', int((d > 2).sum()), ' f64 vs oracle >2:', int((np.minimum(np.abs(f64-want).max(axis=2), 256-np.abs(f64-want).max(axis=2)) > 2).sum()), plan.info())
ys, xs = np.nonzero(d > 2)
for y, x in list(zip(ys, xs))[:12]:
    print((y, x), 'tile', (y // 32, x // 32), 'in-tile', (y % 32, x % 32), 'got', got[y, x], 'f64', f64[y, x], 'oracle', want[y, x], 'il', idx[0, y, x], 'ir', idx[1, y, x], 'w', w[0, y, x], w[1, y, x])
h, w = case.src[1], case.src[2]
for eye in ('l', 'r'):
    fr2 = frame.copy()
    if eye == 'l': fr2[:, w // 2:] = 0
    else: fr2[:, :w // 2] = 0
    dev2 = torch.from_numpy(fr2).cuda()
    g2 = plan.remap(dev2, interpolation='bilinear').cpu().numpy().astype(np.int16)
    plan.set_mode(nat.MODE_FAITHFUL); f2 = plan.remap(dev2, interpolation='bilinear').cpu().numpy().astype(np.int16); plan.set_mode(nat.MODE_AUTO)
    dd = np.abs(g2 - f2).max(axis=2)
    print('eye', eye, 'only: pixels > 2 LSB', int((dd > 2).sum()))
    for y, x in list(zip(*np.nonzero(dd > 2)))[:6]:
        print('   ', (y, x), 'tiles', g2[y, x], 'f64', f2[y, x])
