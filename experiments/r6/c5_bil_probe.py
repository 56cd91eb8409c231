"""Bilinear / nearest kernel time of full-size cases on a cold pool (HIP events), with the bilinear mode's tile mix and launch shape.
    python experiments/r6/c5_bil_probe.py [case ...] [--reps N] [--json]      (cases of tests/cases.py: c1 c2 c3 c5_180 c5_195)"""
import sys, json, torch
sys.path.insert(0, '.')
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import full_cases
names = [a for a in sys.argv[1:] if not a.startswith('-') and not a.isdigit()] or ['c5_180']
reps = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 40
modes = ('bilinear',) if '--bil-only' in sys.argv else ('bilinear', 'nearest')
cases = {c.name: c for c in full_cases()}
for name in names:
    case = cases[name]
    plan = H.pb_plan_private(case)
    _, h, w, *_ = case.src
    n = max(2, (1280 << 20) // (3 * (h * w + case.dst[1] * case.dst[2])) + 1)
    frames = [nat.synth_frame(h, w, frame=f, circle_mask=case.mask) for f in range(n)]
    outs = [torch.empty((case.dst[1], case.dst[2], 3), dtype=torch.uint8, device='cuda') for _ in range(n)]
    res = {}
    for mode in modes:
        kw = {'interpolation': 'bilinear'} if mode == 'bilinear' else {}
        for i in range(2 * n): plan.remap(frames[i % n], outs[i % n], **kw)
        torch.cuda.synchronize()
        ts = []
        for rep in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(8): plan.remap(frames[(8 * rep + i) % n], outs[(8 * rep + i) % n], **kw)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 8)
        ts.sort()
        res[mode] = [round(ts[0], 2), round(ts[len(ts) // 10], 2), round(ts[len(ts) // 2], 2)]
    rec = {'case': name, 'lib': nat.LIB_PATH.split('/')[-1], 'us_min_p10_med': res, 'mix': plan.bilinear_tile_mix(), 'shape': plan.bilinear_launch_shape()}
    print(json.dumps(rec) if '--json' in sys.argv else f"{name} {rec['lib']} {res} mix {rec['mix']} shape {rec['shape']}", flush=True)
