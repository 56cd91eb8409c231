"""ms per frame of the FAITHFUL float64 kernel and warm plan preparation for the BASELINE geometries, with one build of the library:
    python experiments/faithful_time.py <lib.so | -> [config ...]
(round 3: what the correctly rounded sin / cos / atan2 cost, and what the two-step evaluation gives back)"""
import os, sys, time
lib = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if lib != '-':
    os.environ['PB_LIB_PATH'] = os.path.abspath(lib)
import torch
import bench
from photonbend_amd import _native as nat
L = nat.load()
for name in (sys.argv[2:] or ['c1', 'c2', 'c3', 'c5']):
    cfg = bench.CONFIGS[name]
    d, rots, s = bench.build_projs(cfg)
    plan = nat.Plan(d, rots, s)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        p2 = nat.Plan(d, rots, s)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        del p2
    src = nat.synth_frame(s.height, s.width, frame=1, seed=0, circle_mask=cfg['mask'])
    out = torch.empty((d.height, d.width, 3), dtype=torch.uint8, device='cuda')
    plan.set_mode(nat.MODE_FAITHFUL)
    st = nat.current_stream()
    for _ in range(2): nat.check(L.pb_remap_u8(plan.handle, src.data_ptr(), out.data_ptr(), 1, 0, 0, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): nat.check(L.pb_remap_u8(plan.handle, src.data_ptr(), out.data_ptr(), 1, 0, 0, st))
    e1.record(); torch.cuda.synchronize()
    print('%-24s %-4s faithful kernel %8.1f us/frame   warm plan preparation %6.2f ms (min of 5; median %6.2f)' % (
        os.path.basename(lib), name, e0.elapsed_time(e1) * 1e3 / 8, min(ts), sorted(ts)[2]), flush=True)
    del plan, src, out
    torch.cuda.empty_cache()
