import sys, os, ctypes, torch
sys.path.insert(0, '.')
import photonbend_amd.build as b
b.LIB_PATH = os.path.abspath('experiments/libpb_stamps.so')
import photonbend_amd._native as nat
nat.LIB_PATH = b.LIB_PATH
from tests import helpers as H
from tests.cases import full_cases
case = [c for c in full_cases() if c.name == sys.argv[1]][0]
plan = H.pb_plan(case)
lib = nat.load()
_, h, w, *_ = case.src
frames = [nat.synth_frame(h, w, frame=f, circle_mask=case.mask) for f in range(4)]
outs = [torch.empty((case.dst[1], case.dst[2], 3), dtype=torch.uint8, device='cuda') for _ in range(4)]
for i in range(3): plan.remap(frames[i], outs[i])
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
lib.pb_debug_stamps(buf, 1)
N = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(N): plan.remap(frames[i % 4], outs[i % 4])
e1.record(); torch.cuda.synchronize()
lib.pb_debug_stamps(buf, 0)
info = plan.info()
waves = info['lean_tiles'] * N
print('us/frame %.1f' % (e0.elapsed_time(e1) * 1e3 / N))
names = ['issue window loads', 'model math', 'wait loads landed', 'gather (LDS) + issue stores', 'wait stores']
for n, v in zip(names, buf): print('  %-30s %8.0f cycles/wave' % (n, v / waves))
print('  whole tile: lean %.0f cyc x %d, direct %.0f cyc x %d, other(black+generic) %.0f cyc x %d' % (buf[5] / max(1, info['lean_tiles'] * N), info['lean_tiles'], buf[6] / max(1, info['direct_tiles'] * N), info['direct_tiles'], buf[7] / max(1, (info['tiles'] - info['lean_tiles'] - info['direct_tiles'] - info['fix_tiles']) * N), info['tiles'] - info['lean_tiles'] - info['direct_tiles'] - info['fix_tiles']))
