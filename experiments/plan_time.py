"""Where plan creation spends its time: cold / warm creation per BASELINE config, and the split host vs kernels."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import full_cases
torch.cuda.init(); torch.zeros(1, device='cuda'); torch.cuda.synchronize()
for case in full_cases():
    src, cmap = H.pb_chain(case, image=np.zeros((case.src[1], case.src[2], 3), np.uint8))
    d, r, s = cmap.dst_proj, cmap.rotations, src._proj()
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p = nat.Plan(d, r, s)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        tm = p.timing()
        t1 = time.perf_counter(); blob = p.serialize(); t2 = time.perf_counter(); q = nat.Plan.deserialize(blob, d, r, s); torch.cuda.synchronize(); t3 = time.perf_counter()
        del p, q
    t0 = time.perf_counter(); p = nat.Plan(d, r, s, defer=True); t_def = (time.perf_counter() - t0) * 1e3
    print('%-7s create ms %s (library timer %.2f); deferred %.3f ms; serialize %.1f ms (%d KB), deserialize %.1f ms' % (case.name, ['%.2f' % t for t in ts], tm['prepare_ms'], t_def, (t2 - t1) * 1e3, len(blob) // 1024, (t3 - t2) * 1e3))
