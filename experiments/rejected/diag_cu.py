"""A/B of the two hot launches: one wave per tile (MODE_FAST_WAVES) vs CU-resident pipelined waves (MODE_FAST_CU).
usage: python experiments/diag_cu.py c2,c3,c1 [frames_per_launch]"""
import sys, time, torch
sys.path.insert(0, '.')
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import full_cases
names = sys.argv[1].split(',') if len(sys.argv) > 1 else ['c2']
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 1
POOL = 6
for case in full_cases():
    if case.name not in names: continue
    plan = H.pb_plan(case)
    print(case.name, plan.info(), flush=True)
    _, h, w, *_ = case.src
    frames = [torch.stack([nat.synth_frame(h, w, frame=p * nf + f, circle_mask=case.mask) for f in range(nf)]) for p in range(POOL)]
    outs = [torch.empty((nf, case.dst[1], case.dst[2], 3), dtype=torch.uint8, device='cuda') for _ in range(POOL)]
    ref = None
    for mode, name in ((nat.MODE_FAITHFUL, 'faithful'), (nat.MODE_FAST_NARROW, "narrow"), (nat.MODE_FAST, "wide"), (nat.MODE_FAST_NARROW, "narrow"), (nat.MODE_FAST, "wide")):
        plan.set_mode(mode)
        for o in outs: o.fill_(0x5A)
        for i in range(POOL): plan.remap(frames[i], outs[i])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        N = 60
        e0.record()
        for i in range(N): plan.remap(frames[i % POOL], outs[i % POOL])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / N
        if ref is None:
            ref = [o.clone() for o in outs]
            print('   %-9s %8.1f us/launch' % (name, us), flush=True)
        else:
            nd = sum(int((a != b).sum()) for a, b in zip(ref, outs))
            print('   %-9s %8.1f us/launch   differing bytes vs faithful: %d' % (name, us, nd), flush=True)
