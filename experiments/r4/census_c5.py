"""Tile classes of the two eyes of c5's plan as the BILINEAR mode sees them (which tiles take the coordinate table, and why).
Needs the diagnostic build: PB_LIB_PATH=build/libphotonbend_hip_diag.so python experiments/r4/census_c5.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from photonbend_amd import _native as nat
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c5"]
d, rots, s = bench.build_projs(cfg)
plan = nat.Plan(d, rots, s)
info = plan.info(); n = info["tiles"]
lib = nat.load()
tabs = []
for fn in ("pb_debug_copy_table", "pb_debug_copy_table_r"):
    f = getattr(lib, fn); f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    buf = np.zeros((n, 64), np.int32)
    assert f(plan.handle, buf.ctypes.data, buf.nbytes) == 0
    tabs.append(buf)
def cls(t):
    fl, bil = t[:, 2], t[:, 63]
    black = (fl & 8) != 0
    failed = (fl & 2) != 0
    plain = (fl & (4 | 16)) != 0
    out = np.full(len(t), "?", dtype=object)
    out[black] = "black"
    out[~black & (bil < 0) & ((fl & 4) != 0)] = "lean"
    out[~black & (bil < 0) & ((fl & 16) != 0)] = "direct"
    tb = ~black & (bil >= 0)
    out[tb & failed] = "table:failed"
    out[tb & ~failed & ~plain] = "table:generic"
    out[tb & ~failed & plain & ((fl & 1024) != 0)] = "table:masked"
    out[tb & ~failed & plain & ((fl & 1024) == 0) & ((fl & 4096) != 0)] = "table:coarse"
    out[tb & ~failed & plain & ((fl & (1024 | 4096)) == 0)] = "table:rim-reach"
    return out
L, R = cls(tabs[0]), cls(tabs[1])
two = (L != "black") & (R != "black")
print("tiles", n, "two-eye", int(two.sum()), "one-eye", int(((L != "black") ^ (R != "black")).sum()), "both black", int(((L == "black") & (R == "black")).sum()))
from collections import Counter
print("two-eye tiles, (left, right) classes:")
for k, v in Counter(zip(L[two], R[two])).most_common(): print("   ", k, v)
print("one-eye tiles, class of the live eye:")
live = np.where(L != "black", L, R)[~two & ((L != "black") | (R != "black"))]
for k, v in Counter(live).most_common(): print("   ", k, v)
