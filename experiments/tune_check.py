import sys, numpy as np, torch, time
sys.path.insert(0, '.')
import photonbend_amd._native as nat
from tests import helpers as H
from tests.cases import full_cases
for name in ['c2','c1','c3','c5_180']:
    case = [c for c in full_cases() if c.name == name][0]
    src, cmap = H.pb_chain(case, image=np.zeros((case.src[1], case.src[2], 3), np.uint8))
    d, rots, s = cmap.dst_proj, cmap.rotations, src._proj()
    plain = nat.Plan(d, rots, s)
    t0=time.time(); tuned = nat.Plan(d, rots, s, tune=True); dt=time.time()-t0
    _, h, w, *_ = case.src
    f = nat.synth_frame(h, w, frame=3, circle_mask=case.mask)
    a = plain.remap(f).clone(); b = tuned.remap(f)
    blob = tuned.serialize(); twin = nat.Plan.deserialize(blob, d, rots, s); c = twin.remap(f)
    print(name, 'tuned budget', tuned.info()['window_budget'], 'tune_ms %.1f'%tuned.timing()['tune_ms'], 'wall %.2fs'%dt, 'equal', bool(torch.equal(a,b)), bool(torch.equal(a,c)))
