import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import full_cases
names = sys.argv[1].split(',') if len(sys.argv) > 1 else ['c2']
for case in full_cases():
    if case.name not in names: continue
    t0 = time.perf_counter(); plan = H.pb_plan(case); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(case.name, 'plan create %.1f ms' % ((t1 - t0) * 1e3), plan.info())
    _, h, w, *_ = case.src
    frames = [nat.synth_frame(h, w, frame=f, circle_mask=case.mask) for f in range(4)]
    outs = [torch.empty((case.dst[1], case.dst[2], 3), dtype=torch.uint8, device='cuda') for _ in range(4)]
    for mode, name in ((nat.MODE_FAITHFUL, 'faithful'), (nat.MODE_FAST, 'fast'), (nat.MODE_AUTO, 'auto')):
        plan.set_mode(mode)
        for i in range(3): plan.remap(frames[i % 4], outs[i % 4])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        N = 20
        for i in range(N): plan.remap(frames[i % 4], outs[i % 4])
        e1.record(); torch.cuda.synchronize()
        print('   %-9s %.1f us/frame' % (name, e0.elapsed_time(e1) * 1e3 / N), plan.info()['fast_path'])
        if mode == nat.MODE_FAITHFUL: ref = [o.clone() for o in outs]
        else: print('      equal to faithful:', all(torch.equal(a, b) for a, b in zip(ref, outs)))
    for mode, name in ((nat.MODE_FAITHFUL, 'faithful'), (nat.MODE_FAST, 'fast')):
        plan.set_mode(mode)
        for i in range(2): plan.index_map()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(10): idx = plan.index_map()
        e1.record(); torch.cuda.synchronize()
        print('   index-map only %-9s %.1f us' % (name, e0.elapsed_time(e1) * 1e3 / 10))
