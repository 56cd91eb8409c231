"""Does a launch's duration depend on WHICH buffers it runs on?  (c3's p10/p90 are 31-38 us within one bench run.)
    python experiments/addr_effect.py c3"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import photonbend_amd._native as nat
from tests import helpers as H
from tests.cases import full_cases
case = [c for c in full_cases() if c.name == sys.argv[1]][0]
src, cmap = H.pb_chain(case, image=np.zeros((case.src[1], case.src[2], 3), np.uint8))
plan = nat.Plan(cmap.dst_proj, cmap.rotations, src._proj())
_, h, w, *_ = case.src
n = 6
frames = [nat.synth_frame(h, w, frame=f, circle_mask=case.mask) for f in range(n)]
outs = [torch.empty((case.dst[1], case.dst[2], 3), dtype=torch.uint8, device='cuda') for _ in range(n)]
for i in range(n): plan.remap(frames[i], outs[i])
torch.cuda.synchronize()
print('src addrs', [hex(f.data_ptr()) for f in frames]); print('dst addrs', [hex(o.data_ptr()) for o in outs])
res = np.zeros((n, n))
for rep in range(12):
    for i in range(n):
        for j in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); plan.remap(frames[i], outs[j]); plan.remap(frames[i], outs[j]); plan.remap(frames[i], outs[j]); e1.record(); torch.cuda.synchronize()
            if rep >= 2: res[i, j] += e0.elapsed_time(e1) * 1e3 / 3 / 10
np.set_printoptions(precision=1, suppress=True, linewidth=200)
print('us per launch, rows = source frame, columns = destination frame'); print(res)
# same pair repeatedly vs alternating pairs
for name, seq in (('same pair', [(0, 0)] * 60), ('round robin', [(k % n, k % n) for k in range(60)])):
    ts = []
    for (i, j) in seq:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.remap(frames[i], outs[j]); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print(name, 'mean %.1f p10 %.1f p90 %.1f' % (np.mean(ts[10:]), np.percentile(ts[10:], 10), np.percentile(ts[10:], 90)))
