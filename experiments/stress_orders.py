"""Random mid-size geometries (many with grids that divide into super-tiles, so that every launch-order rule gets used):
the fast path's bytes against the faithful float64 kernel's.   python experiments/stress_orders.py [seed] [count]"""
import sys, random, numpy as np, torch
sys.path.insert(0, '.')
import photonbend_amd._native as nat
from tests import helpers as H
from tests.cases import Case, cam, dbl, pano, inscribed
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rng = random.Random(seed)
lenses = ["equidistant", "equisolid", "stereographic", "orthographic", "thoby"]
def side(): return rng.choice([1024, 1280, 1536, 2048, 2304, 2560, 3072]) if rng.random() < 0.8 else rng.randrange(700, 2600)
def fov(l): return {"orthographic": rng.uniform(100, 178), "stereographic": rng.uniform(120, 300), "thoby": rng.uniform(120, 200)}.get(l, rng.uniform(120, 360))
def end(role):
    k = rng.random()
    if k < 0.4:
        hgt = side() // 2 * 2; return pano(hgt, 2 * hgt)
    if k < 0.85 or role == 'dst':
        s = side(); l = rng.choice(lenses); return cam(s, s, l, fov(l), inscribed(s))
    hgt = side() // 2 * 2; return dbl(hgt, 2 * hgt, "equidistant", rng.uniform(180, 200))
bad = 0
for n in range(count):
    d, s = end('dst'), end('src')
    rots = [(rng.uniform(-40, 40), rng.uniform(-180, 180), rng.uniform(-30, 30))] if rng.random() < 0.7 else []
    case = Case(f"s{seed}_{n}", d, s, rots, mask=2 if s[0] == 'double' else (1 if s[0] == 'camera' else 0))
    src, cmap = H.pb_chain(case, image=np.zeros((case.src[1], case.src[2], 3), np.uint8))
    plan = nat.Plan(cmap.dst_proj, cmap.rotations, src._proj())
    f = nat.synth_frame(case.src[1], case.src[2], frame=n, circle_mask=case.mask)
    fast = plan.remap(f).clone()
    fb = plan.remap(torch.stack([f, f]))  # the batch launch
    info = plan.info()
    plan.set_mode(nat.MODE_FAITHFUL)
    ref = plan.remap(f)
    diff = (fast != ref).any(dim=2)
    nd = int(diff.sum().item())
    maxd = int((fast.to(torch.int16) - ref.to(torch.int16)).abs().max().item()) if nd else 0
    okb = bool(torch.equal(fb[0], fast) and torch.equal(fb[1], fast))
    tol = s[0] == 'double' and len(rots) > 0  # rotated double sources: <= 1 LSB in a few pixels (documented tolerance)
    ok = okb and (nd == 0 or (tol and maxd <= 1 and nd <= diff.numel() // 1000))
    bad += not ok
    print('%-8s dst %-28s src %-30s rot %d  fast %s tiles %d lean %d direct %d fail %d: differing px %d (max %d) batch_equal %s %s' % (
        case.name, str(d[:3]) + str(d[3])[:6] if len(d) > 3 else str(d[:3]), str(s[:3]) + (str(s[3])[:6] if len(s) > 3 else ''), len(rots), info['fast_path'], info['tiles'], info['lean_tiles'], info['direct_tiles'], info['fix_tiles'], nd, maxd, okb, 'OK' if ok else 'MISMATCH'), flush=True)
    del plan
print('mismatching cases:', bad)
sys.exit(1 if bad else 0)
