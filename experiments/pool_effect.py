"""Cold launch duration per pool slot: launches rotate over a pool larger than the memory-side cache, one event pair each.
    python experiments/pool_effect.py c3 [pool]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import photonbend_amd._native as nat
from tests import helpers as H
from tests.cases import full_cases
case = [c for c in full_cases() if c.name == sys.argv[1]][0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
src, cmap = H.pb_chain(case, image=np.zeros((case.src[1], case.src[2], 3), np.uint8))
plan = nat.Plan(cmap.dst_proj, cmap.rotations, src._proj())
_, h, w, *_ = case.src
srcs = torch.empty((n, h, w, 3), dtype=torch.uint8, device='cuda')
for f in range(n): nat.synth_frame(h, w, frame=f, circle_mask=case.mask, out=srcs[f])
dsts = torch.empty((n, case.dst[1], case.dst[2], 3), dtype=torch.uint8, device='cuda')
print('src base %x stride %x; dst base %x stride %x' % (srcs.data_ptr(), srcs[1].data_ptr() - srcs.data_ptr(), dsts.data_ptr(), dsts[1].data_ptr() - dsts.data_ptr()))
for k in range(3 * n): plan.remap(srcs[k % n], dsts[k % n])
torch.cuda.synchronize()
R = 40
t = np.zeros((R, n))
for r in range(R):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for k in range(n):
        evs[k][0].record(); plan.remap(srcs[k], dsts[k]); evs[k][1].record()
    torch.cuda.synchronize()
    t[r] = [a.elapsed_time(b) * 1e3 for a, b in evs]
np.set_printoptions(precision=1, suppress=True, linewidth=200)
print('median us per pool slot:', np.median(t[5:], axis=0))
print('p10:', np.percentile(t[5:], 10, axis=0)); print('p90:', np.percentile(t[5:], 90, axis=0))
