import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from photonbend_amd import _native as nat
from tests import helpers as H
from tests.cases import full_cases
case = [c for c in full_cases() if c.name == sys.argv[1]][0]
plan = H.pb_plan(case)
info = plan.info(); n = info['tiles']
buf = np.zeros((n, 64), np.int32)
lib = nat.load()
lib.pb_debug_copy_table.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
assert lib.pb_debug_copy_table(plan.handle, buf.ctypes.data, buf.nbytes) == 0
flags = buf[:, 2]; rows = buf[:, 3]; r0 = buf[:, 54]; c0 = buf[:, 55]; cols = buf[:, 56]
lean = (flags & 4) != 0; black = (flags & 8) != 0; failed = (flags & 2) != 0
gen = ~lean & ~black & ~failed
print(info)
print('generic tiles', gen.sum())
tx = np.arange(n) % (case.dst[2] // 32); ty = np.arange(n) // (case.dst[2] // 32)
g = np.where(gen)[0]
print('generic: rows pct', np.percentile(rows[g], [0, 25, 50, 75, 100]), 'cols pct', np.percentile(cols[g], [0, 25, 50, 75, 100]))
bytes_ = rows[g] * ((3 * cols[g] + 32) // 16 * 16)
print('window bytes pct', np.percentile(bytes_, [0, 25, 50, 75, 100]), ' >20224:', (bytes_ > 20224).sum())
rad = np.hypot(tx[g] * 32 + 16 - case.dst[2] / 2, ty[g] * 32 + 16 - case.dst[1] / 2)
print('radius of generic tiles pct', np.percentile(rad, [0, 10, 25, 50, 75, 90, 100]))
print('lean rows/cols median', np.median(rows[lean]), np.median(cols[lean]))
