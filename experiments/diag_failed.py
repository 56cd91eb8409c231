"""Where are a plan's failed tiles?  Prints a coarse map (one char per 4x4 tiles) and radius statistics.
Needs the -DPB_STAMPS build (pb_debug_copy_table): python experiments/diag_failed.py c3"""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, '.')
import photonbend_amd.build as b
b.LIB_PATH = os.path.abspath('experiments/libpb_stamps.so')
import photonbend_amd._native as nat
nat.LIB_PATH = b.LIB_PATH
from tests import helpers as H
from tests.cases import full_cases
case = [c for c in full_cases() if c.name == sys.argv[1]][0]
plan = H.pb_plan(case)
info = plan.info(); n = info['tiles']
buf = np.zeros((n, 64), np.int32)
lib = nat.load()
lib.pb_debug_copy_table.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
assert lib.pb_debug_copy_table(plan.handle, buf.ctypes.data, buf.nbytes) == 0
flags = buf[:, 2]
ntx = case.dst[2] // 32; nty = case.dst[1] // 32
failed = ((flags & 2) != 0).reshape(nty, ntx)
print(info)
tx = np.arange(n) % ntx; ty = np.arange(n) // ntx
rad = np.hypot(tx * 32 + 16 - case.dst[2] / 2, ty * 32 + 16 - case.dst[1] / 2)
f = np.where(failed.reshape(-1))[0]
print('failed', len(f), 'radius pct', np.percentile(rad[f], [0, 10, 25, 50, 75, 90, 100]))
print('failed with radius > 1950:', (rad[f] > 1950).sum(), ' < 1950:', (rad[f] <= 1950).sum())
for y in range(0, nty, 4):
    print(''.join('#' if failed[y:y+4, x:x+4].any() else '.' for x in range(0, ntx, 4)))
