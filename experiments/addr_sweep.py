"""Launch duration against the BASE ADDRESS of the destination (and source) frame: offsets inside one large allocation.
    python experiments/addr_sweep.py c2"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import photonbend_amd._native as nat
from tests import helpers as H
from tests.cases import full_cases
case = [c for c in full_cases() if c.name == sys.argv[1]][0]
src, cmap = H.pb_chain(case, image=np.zeros((case.src[1], case.src[2], 3), np.uint8))
plan = nat.Plan(cmap.dst_proj, cmap.rotations, src._proj())
_, h, w, *_ = case.src
dh, dw = case.dst[1], case.dst[2]
sb, db = 3 * h * w, 3 * dh * dw
n = 6
SPAN = 80 << 20
big_src = torch.empty(n * (sb + SPAN), dtype=torch.uint8, device='cuda'); big_src.random_(0, 255)
big_dst = torch.empty(n * (db + SPAN), dtype=torch.uint8, device='cuda')
print('big_src %x big_dst %x' % (big_src.data_ptr(), big_dst.data_ptr()))
def view(big, k, off, nbytes, shape): 
    o = k * (nbytes + SPAN) + off
    return big[o:o + nbytes].view(shape)
def timeit(soff, doff, reps=5):
    ts = []
    for rep in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(n): plan.remap(view(big_src, k, soff, sb, (h, w, 3)), view(big_dst, k, doff, db, (dh, dw, 3)))
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return np.median(ts[1:])
offs = [0, 256, 1024, 4096, 16384, 65536, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 6 << 20, 8 << 20, 16 << 20, 32 << 20, 48 << 20, 64 << 20]
base = timeit(0, 0)
print('dst offset sweep (src offset 0):')
for o in offs: print('  %9d  %.2f us' % (o, timeit(0, o)))
print('src offset sweep (dst offset 0):')
for o in offs: print('  %9d  %.2f us' % (o, timeit(o, 0)))
