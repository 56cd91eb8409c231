"""Nearest-mode kernel time of a full-size case on a cold pool (the diagnostic build's PB_EXP skips tile classes).  python experiments/r6/near_probe.py c5_180"""
import sys, json, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import full_cases
name = sys.argv[1] if len(sys.argv) > 1 else 'c5_180'
case = {c.name: c for c in full_cases()}[name]
plan = H.pb_plan_private(case)
_, h, w, *_ = case.src
n = max(2, (1280 << 20) // (3 * (h * w + case.dst[1] * case.dst[2])) + 1)
frames = [nat.synth_frame(h, w, frame=f, circle_mask=case.mask) for f in range(n)]
outs = [torch.empty((case.dst[1], case.dst[2], 3), dtype=torch.uint8, device='cuda') for _ in range(n)]
for i in range(2 * n): plan.remap(frames[i % n], outs[i % n])
torch.cuda.synchronize()
ts = []
for rep in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(8): plan.remap(frames[(8 * rep + i) % n], outs[(8 * rep + i) % n])
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / 8)
ts.sort()
print(name, __import__("os").environ.get("PB_EXP", "-"), 'nearest us min/p10/median', round(ts[0], 2), round(ts[4], 2), round(ts[20], 2), flush=True)
